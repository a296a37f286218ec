"""Soak: many small batches through the engine (two sampling pipelines, lookahead) against the oracle replay.
    python tools/soak_engine.py [sample_type] [num_epoch] [extra engine config k=v ...]   e.g. pipelines=2 extract_streams=2"""
import os, subprocess, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_engine as te  # noqa: E402

stype = sys.argv[1] if len(sys.argv) > 1 else "khop3"
epochs = int(sys.argv[2]) if len(sys.argv) > 2 else 6
# random walk: tests/engine_driver.py configures PinSAGE's defaults (walk 3, restart 0.5, 4 walks, 5 neighbours per layer)
fan = [5, 5, 5] if stype == "random_walk" else [6, 5, 4]
okw = dict(walk_length=3, restart_prob=0.5, num_walk=4) if stype == "random_walk" else {}
with tempfile.TemporaryDirectory() as tmp:
    import pathlib
    d = te.make_dataset(pathlib.Path(tmp) / "ds", num_node=6000, dim=8, num_train=2000, seed=9)
    prefix = os.path.join(tmp, "out")
    r = subprocess.run([sys.executable, te.DRIVER, d["path"], prefix, "arch1", "1", f"sample_type={stype}", "seed=5",
                        "batch_size=16", f"num_epoch={epochs}", "fanout=" + " ".join(map(str, fan))] + sys.argv[3:], capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-3000:]
    want = te._oracle_batches(d, 0, 1, 16, epochs, fan, 5, arch6=False, sample_type=stype, **okw)
    te._check(np.load(prefix + ".w0.npz"), want, 3)
    print(f"soak ok: {len(want)} batches, sample_type {stype} {' '.join(sys.argv[3:])}")
