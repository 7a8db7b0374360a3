#!/usr/bin/env python3
"""profiles/pmc_traffic.json from the FETCH_SIZE and WRITE_SIZE passes of tools/profile_round.sh, stamped with the sha of
the kernel sources so bench.py reports `roofline.traffic` only for the build the passes were taken on.

    python tools/make_pmc_traffic.py gpurun_out/<tag>/pmc_fetch/*.db gpurun_out/<tag>/pmc_write/*.db <profile-name> [kernel-regex]
"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import bench            # noqa: E402
import rocpd_pmc        # noqa: E402


def per_launch(db, counter, pattern):
    agg = rocpd_pmc.table(db)
    rows = [(k, v) for (k, c), v in agg.items() if c == counter and re.search(pattern, k)]
    n = sum(v[0] for _, v in rows); tot = sum(v[1] for _, v in rows)
    return tot / max(n, 1), n, [k for k, _ in rows]


if __name__ == '__main__':
    fetch_db, write_db, name = sys.argv[1:4]
    pat = sys.argv[4] if len(sys.argv) > 4 else r'conv3x3_wide_kernel<true|conv3x3_band_kernel<4, 2, true'
    f_kb, nf, ks = per_launch(fetch_db, 'FETCH_SIZE', pat)
    w_kb, nw, _ = per_launch(write_db, 'WRITE_SIZE', pat)
    # MI355X_MICROARCH.md section HBM: on gfx950 FETCH_SIZE tallies the 128-B requests of wide (16 B/lane) streaming
    # reads at 64 B -> x2; WRITE_SIZE is taken as reported (it equals h'+c' exactly for this kernel)
    out = {"_comment": "HBM-side traffic of the dominant kernel (ConvLSTM gate convolutions) from rocprofv3 PMC passes; "
                       "see profiles/README.md", "kernel": ks, "profile": name, "launches": [nf, nw],
           "fetch_size_kb_per_launch_raw": f_kb, "fetch_correction": 2.0, "write_size_kb_per_launch": w_kb,
           "convlstm_bytes_per_launch": int((2.0 * f_kb + w_kb) * 1024), "source_sha": bench.source_sha(),
           "arith": "mx6" if any('m6::' in k for k in ks) else ("h3" if any('h3::' in k for k in ks) else "mx")}
    json.dump(out, open(os.path.join(ROOT, 'profiles', 'pmc_traffic.json'), 'w'), indent=1)
    print(json.dumps(out, indent=1))
