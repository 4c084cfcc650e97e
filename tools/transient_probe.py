"""The first steps after an idle GPU: per-step device time of the headline step next to the clocks and board power
the driver reports (sysfs), to find what the 415 -> 537 -> 417 us bump of the backward in steps 5..25 is
(profiles/r01/bench_steps_sweep.txt).  Developer tool.

    python tools/transient_probe.py [steps]
"""
import glob
import os
import sys
import threading
import time

import torch

sys.path.insert(0, '.')


def sysfs_sources(pci_bus_id=None):
    """clock / power files of the card at `pci_bus_id` (all cards if None)"""
    out = {}
    for card in sorted(glob.glob('/sys/class/drm/card*/device')):
        if pci_bus_id and os.path.basename(os.path.realpath(card)).lower() != pci_bus_id.lower():
            continue
        for name in ('pp_dpm_sclk', 'pp_dpm_mclk', 'pp_dpm_fclk', 'pp_dpm_socclk'):
            p = os.path.join(card, name)
            if os.path.exists(p):
                out[os.path.basename(os.path.dirname(card)) + ':' + name] = p
        for hw in glob.glob(os.path.join(card, 'hwmon', 'hwmon*')):
            for name in ('power1_average', 'power1_input', 'freq1_input', 'freq2_input', 'temp1_input', 'temp2_input'):
                p = os.path.join(hw, name)
                if os.path.exists(p):
                    out[os.path.basename(os.path.dirname(card)) + ':' + name] = p
    return out


def read(p):
    try:
        with open(p) as fh:
            t = fh.read().strip()
    except OSError as e:
        return 'ERR %s' % e.errno
    if '\n' in t:  # pp_dpm tables: keep the active level (marked *)
        act = [ln for ln in t.splitlines() if ln.endswith('*')]
        return act[0] if act else t.replace('\n', ' | ')
    return t


class Sampler(threading.Thread):
    def __init__(self, srcs, period=0.010):
        super().__init__(daemon=True)
        self.srcs, self.period, self.rows, self.stop = srcs, period, [], False

    def run(self):
        while not self.stop:
            t = time.perf_counter()
            self.rows.append((t, {k: read(p) for k, p in self.srcs.items()}))
            time.sleep(self.period)


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 80
    from bench import build_quantizer
    dev = torch.device('cuda', 0)
    torch.manual_seed(123456)
    shape = (256, 512, 56, 56)
    x = torch.randn(shape, device=dev, dtype=torch.bfloat16).requires_grad_(True)
    g = torch.randn(shape, device=dev, dtype=torch.bfloat16)
    q = build_quantizer(512, True, dev)
    props = torch.cuda.get_device_properties(0)
    bus = getattr(props, 'pci_bus_id', None)
    dom = getattr(props, 'pci_domain_id', 0)
    devid = getattr(props, 'pci_device_id', 0)
    pci = '%04x:%02x:%02x.0' % (dom, bus, devid) if bus is not None else None
    srcs = sysfs_sources(pci)
    if not srcs:
        print('no sysfs card matches %s; sampling nothing' % pci)
    want = ('pp_dpm_sclk', 'power1_input', 'power1_average', 'freq1_input')
    srcs = {k: p for k, p in srcs.items() if k.split(':', 1)[1] in want and not read(p).startswith('ERR')}
    print('sysfs sources:', sorted(srcs), flush=True)

    def step():
        x.grad = None
        y, _, _, _ = q(x)
        y.backward(g)

    for label, idle, sample in (('after init', 0.0, True), ('after 3 s idle', 3.0, True), ('after 3 s idle, no sampler', 3.0, False),
                               ('after 0.2 s idle, no sampler', 0.2, False)):
        torch.cuda.synchronize()
        time.sleep(idle)
        sm = Sampler(srcs if sample else {})
        sm.start()
        evs = []
        t0 = time.perf_counter()
        for i in range(steps):
            a = torch.cuda.Event(enable_timing=True)
            a.record()
            step()
            evs.append(a)
        last = torch.cuda.Event(enable_timing=True)
        last.record()
        evs.append(last)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        sm.stop = True
        sm.join()
        ms = [evs[i].elapsed_time(evs[i + 1]) for i in range(steps)]
        start = [evs[0].elapsed_time(evs[i]) for i in range(steps)]
        print('== %s: %d steps in %.1f ms host time' % (label, steps, (t1 - t0) * 1e3))
        print('step  start_ms  step_ms')
        for i, (s, m) in enumerate(zip(start, ms)):
            print('%4d  %8.2f  %.4f' % (i, s, m))
        print('-- sysfs samples (ms since first step submitted; changes only)')
        last = None
        for t, row in sm.rows:
            key = tuple(sorted(row.items()))
            if key != last:
                print('%8.2f  %s' % ((t - t0) * 1e3, '  '.join('%s=%s' % (k.split(':', 1)[1], v) for k, v in sorted(row.items()))))
                last = key
        sys.stdout.flush()


if __name__ == '__main__':
    main()
