"""Helpers for the multi-process tests: ranks come from the fork server conftest.py starts while the test process is
still clean (a process that holds the GPU must not start other programs on the box), a stuck rank is terminated
instead of being left behind, and the rendezvous has a timeout of its own."""
import datetime
import multiprocessing as mp
import os
import socket


def free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def init_gloo(rank, world, port, timeout_s=120):
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world, timeout=datetime.timedelta(seconds=timeout_s))


def run_ranks(worker, world, *args, timeout=180):
    """start `world` processes running worker(rank, world, port, *args, queue); every rank must put (rank, 'ok')"""
    ctx = mp.get_context('forkserver')
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=worker, args=(r, world, port) + tuple(args) + (q,)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=timeout)
    stuck = [p for p in procs if p.is_alive()]
    for p in stuck:
        p.terminate()
        p.join(timeout=10)
    assert not stuck, '%d rank(s) did not finish within %d s (terminated)' % (len(stuck), timeout)
    results = [q.get(timeout=5) for _ in range(world)]
    for rank, msg in results:
        assert msg == 'ok', 'rank %d failed:\n%s' % (rank, msg)
    assert all(p.exitcode == 0 for p in procs)
