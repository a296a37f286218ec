"""What FETCH_SIZE / WRITE_SIZE say about the row gather at a given row width: the guide (MI355X_MICROARCH.md, HBM
section) calibrates the counters for wide coalesced STREAMS only ("other access widths are uncalibrated: calibrate on
a known byte count in your own access pattern"; "ratios between variants of one kernel are unaffected").  This driver
launches the SAME gather kernel (ggms_extract, k_gather_rows<16, PlainRows>) four times per row width under
`rocprofv3 --pmc ...`:

    launch 1, 2   index = 0, 1, 2, ... (rows in table order: the reads ARE one contiguous stream of n * row_bytes --
                  the known byte count)
    launch 3, 4   index = a random selection of n of the N table rows (what a batch gathers)

The counters of launch 3, 4 over those of launch 1, 2 are the gather's over-fetch at that row width: whole 64- / 128-byte
requests behind rows that do not start or end on a request boundary (400-byte rows: 4 x 128 B at best).
Used through tools/pmc_gather_calibration.sh, which runs the two passes and prints the ratios.

    python tools/pmc_gather_calibration.py [dims, default 100 128]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xgnn_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)
N, n = 2_449_029, 1_280_000  # products-sized table, one batch's worth of rows
for dim in [int(x) for x in (sys.argv[1:] or ["100", "128"])]:
    feat = torch.randn(N, dim, device=dev)
    out = torch.empty((n, dim), dtype=torch.float32, device=dev)
    seq = torch.arange(n, dtype=torch.int32, device=dev)
    rnd = torch.randperm(N, device=dev)[:n].to(torch.int32)
    # something else between the launches, so that no launch finds the previous one's lines in the 256-MB Infinity Cache
    spoil = torch.empty(1 << 28, dtype=torch.float32, device=dev)  # 1 GiB
    for idx in (seq, seq, rnd, rnd):
        spoil.fill_(1.0)
        ops.extract(feat, idx, out=out)
        torch.cuda.synchronize()
    print(f"dim {dim}: rows {n}, row bytes {dim * 4}, contiguous bytes read {n * dim * 4}", flush=True)
    del feat, out, spoil
