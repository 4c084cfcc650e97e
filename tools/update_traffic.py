#!/usr/bin/env python
"""profiles/traffic.json from a tools/profile.sh run: HBM bytes per launch of the headline step's kernels (separate
rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes; FETCH_SIZE doubled per the gfx950 note of MI355X_MICROARCH.md), stamped
with the digest of the library sources it was measured on -- bench.py prints `traffic: null` when the library it runs
has other sources.  Run right after the profile, before touching brevitas_amd/csrc or include/.

    python tools/update_traffic.py gpurun_out/<dir>/prof 'profiles/r03/rocprofv3_summary.md, round-3 final build'
"""
import json
import os
import sys

sys.path.insert(0, '.')
from brevitas_amd.csrc import build  # noqa: E402


def main():
    prof, source = sys.argv[1], sys.argv[2]
    digest = build.source_digest()
    if digest is None:
        sys.exit('brevitas_amd/libbvq.so is not the build of the sources on disk: rebuild, re-profile, then stamp')
    raw = json.load(open(os.path.join(prof, 'traffic_raw.json')))

    def bytes_of(prefix):
        for k, v in raw.items():
            if prefix in k:
                return int(round((2 * v['FETCH_SIZE'] + v['WRITE_SIZE']) * 1024))
        return None
    out = {
        '_doc': 'HBM bytes per launch from rocprofv3 PMC counters (separate --pmc FETCH_SIZE and --pmc WRITE_SIZE passes of '
                'bench.py). FETCH_SIZE and WRITE_SIZE are reported in KiB; on gfx950 FETCH_SIZE counts a wide coalesced '
                'streaming read at exactly half its bytes (MI355X_MICROARCH.md, HBM section), so bytes = (2*FETCH_SIZE + '
                'WRITE_SIZE)*1024.',
        '_source': source,
        '_source_digest': digest,
        'act_per_channel_bf16': {'bvq_fakequant_bwd': bytes_of('fakequant_bwd_kernel'),
                                 'bvq_fakequant_fwd': bytes_of('fakequant_fwd_kernel'),
                                 'bvq_stats': bytes_of('absmax_onepass_kernel') or bytes_of('absmax_kernel')},
    }
    with open(os.path.join('profiles', 'traffic.json'), 'w') as fh:
        json.dump(out, fh, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
