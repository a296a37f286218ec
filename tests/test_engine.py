"""Engine behind the samgraph_* ABI vs the oracle: arch1 (one GPU) and arch6 (forked workers, GGMS shards)."""
import os
import subprocess
import sys

import numpy as np
import pytest

import oracle
from graphgen import exact_features, powerlaw_csr

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DRIVER = os.path.join(ROOT, "tests", "engine_driver.py")


def make_dataset(tmp_path, num_node=3000, dim=20, num_train=500, dtype=np.float32, seed=5):
    from xgnn_amd import datagen
    ip, ix = powerlaw_csr(num_node, mean_deg=15, seed=seed)
    train = np.random.RandomState(seed).permutation(num_node)[:num_train].astype(np.uint32)
    feat = exact_features(num_node, dim, dtype)
    label = (np.arange(num_node, dtype=np.int64) * 7) % 13
    g = dict(indptr=ip, indices=ix, train_set=train, meta=dict(feat_dim=dim, num_class=13))
    names = {np.dtype(np.float32): "F32", np.dtype(np.float16): "F16", np.dtype(np.uint8): "U8"}
    datagen.write_dataset(str(tmp_path), g, feat=feat, label=label, feat_dtype=names[np.dtype(dtype)])
    return dict(ip=ip, ix=ix, train=train, feat=feat, label=label, path=str(tmp_path))


def test_data_init_host_only(tmp_path):
    """config + data_init touch no GPU (they run in the parent before fork): dataset format round trip."""
    d = make_dataset(tmp_path)
    code = f"""
import sys; sys.path.insert(0, {ROOT!r})
import samgraph.torch as sam
sam.config({{'dataset_path': {d['path']!r}, '_arch': 6, '_sample_type': 7, 'batch_size': 64, 'num_epoch': 1,
  '_cache_policy': 0, 'cache_percentage': 0.3, 'max_sampling_jobs': 1, 'max_copying_jobs': 1, 'omp_thread_num': 1,
  'num_layer': 2, 'num_hidden': 8, 'lr': 0.1, 'dropout': 0.5, 'num_worker': 2, 'num_fanout': 2, 'fanout': [5, 4]}})
sam.data_init()
f = sam.get_dataset_feat(); l = sam.get_dataset_label()
print(sam.num_class(), sam.feat_dim(), tuple(f.shape), float(f[17, 3]), int(l[17]))
"""
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    assert out.stdout.split()[:2] == ["13", "20"]
    assert "(3000, 20)" in out.stdout
    assert float(out.stdout.split()[-2]) == float(d["feat"][17, 3]) and int(out.stdout.split()[-1]) == int(d["label"][17])


def test_bad_config_aborts(tmp_path):
    """A failed CHECK prints and abort()s (logging.cc:69-73): no error codes at this boundary."""
    code = f"""
import sys; sys.path.insert(0, {ROOT!r})
import samgraph.torch as sam
sam.config({{'dataset_path': '/nonexistent', '_arch': 1}})
"""
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert out.returncode < 0  # SIGABRT
    assert "missing config key" in out.stderr


def test_arch6_refuses_workers_whose_gpus_cannot_reach_each_other(tmp_path):
    """PartitionSolver's P2P matrix (cuda/dist_graph.cu:812-818) as the engine uses it: data_init reads the topology
    file (here a hand-written one in the reference's format: two GPUs, GPU 0 cannot access GPU 1) and aborts naming the
    pair BEFORE any worker is forked or any shard built -- host only, no GPU touched.  A fully connected file passes."""
    d = make_dataset(tmp_path / "ds")
    body = "GPU Count 2\nDevice Order test\nGPU [0] gfx950\nGPU [1] gfx950\n\n\nP2P Matrix\n   1    {a} \n   1    1 \n\n\n" \
           "Bandwidth Matrix\n 5000.00    48.10 \n   47.90  5000.00 \n"
    code = f"""
import sys; sys.path.insert(0, {ROOT!r})
import samgraph.torch as sam
sam.config({{'dataset_path': {d['path']!r}, '_arch': 6, '_sample_type': 7, 'batch_size': 64, 'num_epoch': 1,
  '_cache_policy': 0, 'cache_percentage': 0.3, 'max_sampling_jobs': 1, 'max_copying_jobs': 1, 'omp_thread_num': 1,
  'num_layer': 2, 'num_hidden': 8, 'lr': 0.1, 'dropout': 0.5, 'num_worker': 2, 'num_fanout': 2, 'fanout': [5, 4],
  'part_cache': 'True', 'gpu_extract': 'True'}})
sam.data_init()
print('placed')
"""
    env = {k: v for k, v in os.environ.items() if k != "SAMGRAPH_FORCE_DEVICE"}
    for a, ok in (("0", False), ("1", True)):
        topo = tmp_path / f"topo_{a}"
        topo.write_text(body.format(a=a))
        out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300,
                             env=dict(env, SAMGRAPH_TOPO_FILE=str(topo), SAMGRAPH_LOG_LEVEL="info"))
        if ok:
            assert out.returncode == 0 and "placed" in out.stdout, out.stderr[-1500:]
            assert "GB/s INTO row FROM column" in out.stderr and "48.1" in out.stderr  # the matrix is logged (info level)
        else:
            assert out.returncode < 0 and "placed" not in out.stdout  # SIGABRT, like a failed CHECK of the reference
            assert "GPU 0 cannot access GPU 1" in out.stderr and "hipDeviceCanAccessPeer" in out.stderr


def _replay_arch0(d, batch_size, num_epoch, fanouts, seed, sample_type, sampler, extract):
    """CPUEngine semantics (cpu_loops.cc:41-228): CPUShuffler (no padding, shuffled in place epoch after epoch), then
    per batch the CPU sampler + first-occurrence dedup/remap + CPUExtract.  `sampler` / `extract` are the leaves under
    test: the oracle's ports or the reference's own objects (oracle/_ref); the caller resets their RNG first."""
    data = d["train"].copy()
    n_step = (data.size + batch_size - 1) // batch_size
    ix = d["ix"].copy()  # khop2 permutes the lists
    out = {}
    for ep in range(num_epoch):
        data = oracle.shuffle_minstd0(data, seed + ep)
        for st in range(n_step):
            seeds = data[st * batch_size:(st + 1) * batch_size]
            ht = oracle.HashTable(d["ip"].size - 1, oracle.predict_num_nodes(seeds.size, fanouts, len(fanouts)) + 1)
            ht.fill_with_duplicates(seeds)
            cur, layers = seeds, [None] * len(fanouts)
            for i in range(len(fanouts) - 1, -1, -1):
                src, dst = sampler(d["ip"], ix, cur, fanouts[i])
                ht.fill_with_duplicates(dst)
                col, row = ht.map_edges(src, dst)
                layers[i] = dict(row=row, col=col, num_src=ht.num_items, num_dst=cur.size, data=None)
                cur = ht.unique()
            out[ep * n_step + st] = dict(res=dict(layers=layers, input_nodes=cur), seeds=seeds,
                                         feat=extract(d["feat"], cur), label=d["label"][seeds])
    return out


@pytest.mark.parametrize("sample_type", ["khop0", "khop2"])
def test_arch0_cpu_engine_end_to_end(tmp_path, sample_type):
    """BASELINE configs[0]: the CPU deployment (cpu_engine.cc) through the samgraph_* ABI, trainer on the host, one
    sampling thread -- against the oracle's ports AND, where oracle/_ref is built, against the reference's own
    CPUSampleKHop0/2 + CPUExtract objects run in the same call order (default-seeded thread_local mt19937)."""
    d = make_dataset(tmp_path / "ds")
    prefix = str(tmp_path / "out")
    r = subprocess.run([sys.executable, DRIVER, d["path"], prefix, "arch0", "1", f"sample_type={sample_type}", "seed=7",
                        "batch_size=64", "fanout=5 4", "trainer_ctx=cpu:0", "omp_thread_num=1", "num_epoch=2"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    got = np.load(prefix + ".w0.npz")
    port = {"khop0": oracle.cpu_sample_khop0, "khop2": oracle.cpu_sample_khop2}[sample_type]
    oracle.cpu_random_reset()
    _check(got, _replay_arch0(d, 64, 2, [5, 4], 7, sample_type, port, oracle.extract), 2)
    if oracle.ref_lib() is not None:  # the reference's objects keep their RNG per process: replay in a fresh one
        code = f"""
import sys, pickle; sys.path.insert(0, {ROOT!r}); sys.path.insert(0, {os.path.join(ROOT, 'tests')!r})
import numpy as np, oracle, test_engine
d = pickle.load(open({str(tmp_path / 'd.pkl')!r}, 'rb'))
ref = {{'khop0': oracle.ref_cpu_sample_khop0, 'khop2': oracle.ref_cpu_sample_khop2}}[{sample_type!r}]
want = test_engine._replay_arch0(d, 64, 2, [5, 4], 7, {sample_type!r}, ref, oracle.ref_cpu_extract)
test_engine._check(np.load({prefix + '.w0.npz'!r}), want, 2)
print('ref-ok')
"""
        import pickle
        pickle.dump(d, open(str(tmp_path / "d.pkl"), "wb"))
        rr = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
        assert rr.returncode == 0 and "ref-ok" in rr.stdout, rr.stderr[-2000:]


@pytest.mark.gpu
@pytest.mark.parametrize("pipelined", [0, 1])
def test_arch0_gpu_trainer(tmp_path, pipelined):
    """arch0 as the reference runs it: CPU sampler + extractor, batch copied to the trainer GPU (DoGraphCopy /
    DoFeatureCopy, cpu_loops.cc:230-299); also through the background thread (extract_start)."""
    d = make_dataset(tmp_path / "ds")
    prefix = str(tmp_path / "out")
    r = subprocess.run([sys.executable, DRIVER, d["path"], prefix, "arch0", "1", "sample_type=khop0", "seed=7",
                        "batch_size=64", "fanout=5 4", "trainer_ctx=cuda:0", "omp_thread_num=1", "num_epoch=2",
                        f"pipelined={pipelined}"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    oracle.cpu_random_reset()
    _check(np.load(prefix + ".w0.npz"), _replay_arch0(d, 64, 2, [5, 4], 7, "khop0", oracle.cpu_sample_khop0, oracle.extract), 2)


def test_arch0_empty_feat_mock_table(tmp_path):
    """SAMGRAPH_EMPTY_FEAT = k (engine.cc:198-235, cpu_extraction.cc:47-62): the feature table is a 2^k-row stand-in
    and node v reads row v & (2^k - 1); ours starts as the first 2^k rows of feat.bin, so the rows can be checked."""
    d = make_dataset(tmp_path / "ds")
    prefix = str(tmp_path / "out")
    r = subprocess.run([sys.executable, DRIVER, d["path"], prefix, "arch0", "1", "sample_type=khop0", "seed=7",
                        "batch_size=64", "fanout=5 4", "trainer_ctx=cpu:0", "omp_thread_num=1", "num_epoch=1"],
                       capture_output=True, text=True, timeout=600, env=dict(os.environ, SAMGRAPH_EMPTY_FEAT="5"))
    assert r.returncode == 0, r.stderr[-2000:]
    got = np.load(prefix + ".w0.npz")
    for key in sorted({int(k.split(":")[0]) for k in got.files}):
        inp = got[f"{key}:input_nodes"].view(np.uint32)
        assert got[f"{key}:feat"].tobytes() == d["feat"][inp & 31].astype(np.float32).tobytes()


@pytest.mark.gpu
def test_arch1_debug_aids(tmp_path):
    """SAMGRAPH_EMPTY_FEAT (GPUMockExtract), SAMGRAPH_SANITY_CHECK (per-batch checks) and the node access report on
    the GPU engine; sampling is unaffected (same COO as the oracle replay)."""
    d = make_dataset(tmp_path / "ds")
    prefix = str(tmp_path / "out")
    env = dict(os.environ, SAMGRAPH_EMPTY_FEAT="6", SAMGRAPH_SANITY_CHECK="1", SAMGRAPH_LOG_NODE_ACCESS_SIMPLE="1")
    code = f"""
import sys, os; sys.path.insert(0, {ROOT!r}); os.chdir({str(tmp_path)!r})
import numpy as np, samgraph.torch as sam
sam.config({{'dataset_path': {d['path']!r}, '_arch': 1, '_sample_type': 7, 'batch_size': 64, 'num_epoch': 2,
  '_cache_policy': 0, 'cache_percentage': 0.0, 'max_sampling_jobs': 1, 'max_copying_jobs': 1, 'omp_thread_num': 1,
  'num_layer': 2, 'num_hidden': 8, 'lr': 0.1, 'dropout': 0.5, 'sampler_ctx': 'cuda:0', 'trainer_ctx': 'cuda:0',
  'num_fanout': 2, 'fanout': [5, 4], 'seed': 3}})
sam.init()
visits = np.zeros({d['ip'].size - 1}, np.int64)
ok = True
for _ in range(sam.num_epoch() * sam.num_local_step()):
    sam.sample_once(); key = sam.get_next_batch()
    inp = sam.get_graph_input_nodes(key).cpu().numpy().view(np.uint32)
    feat = sam.get_graph_feat(key).cpu().numpy()
    full = np.fromfile(os.path.join({d['path']!r}, 'feat.bin'), np.float32).reshape(-1, feat.shape[1])
    ok = ok and feat.tobytes() == full[inp & 63].tobytes()
    visits[inp] += 1
sam.report_node_access()
sam.shutdown()
import glob
ranked = np.fromfile(glob.glob('node_access_optimal_cache_bin*.txt')[0], np.uint32)
freq = np.fromfile(glob.glob('node_access_optimal_cache_freq_bin*.txt')[0], np.float32)
ok = ok and ranked.size == visits.size and np.array_equal(freq * 2, visits[ranked].astype(np.float32))
ok = ok and (np.diff(freq) <= 0).all() and len(open(glob.glob('node_access_optimal_cache_hit*.txt')[0]).read().splitlines()) == 101
print('aids-ok' if ok else 'aids-bad')
"""
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "aids-ok" in r.stdout, (r.stdout[-500:], r.stderr[-2000:])


def test_arch0_threads_keep_the_structure(tmp_path):
    """Four sampling threads (static blocks, a generator per thread): draws differ from the one-thread run, the
    structure may not -- every seed keeps min(deg, fanout) distinct neighbours of its own list, ids are dense, rows
    are the right feature rows."""
    d = make_dataset(tmp_path / "ds")
    prefix = str(tmp_path / "out")
    r = subprocess.run([sys.executable, DRIVER, d["path"], prefix, "arch0", "1", "sample_type=khop0", "seed=7",
                        "batch_size=64", "fanout=5 4", "trainer_ctx=cpu:0", "omp_thread_num=4", "num_epoch=1"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    got = np.load(prefix + ".w0.npz")
    ip, ix = d["ip"], d["ix"]
    deg = ip[1:].astype(np.int64) - ip[:-1]
    keys = sorted({int(k.split(":")[0]) for k in got.files})
    assert len(keys) == (d["train"].size + 63) // 64
    for key in keys:
        inp = got[f"{key}:input_nodes"].view(np.uint32)
        assert np.unique(inp).size == inp.size
        for i, f in ((1, 4), (0, 5)):
            row, col = got[f"{key}:row{i}"].view(np.uint32), got[f"{key}:col{i}"].view(np.uint32)
            nd = int(got[f"{key}:num_dst{i}"])
            assert np.array_equal(np.bincount(col, minlength=nd), np.minimum(deg[inp[:nd]], f))
            for e in range(0, row.size, 7):
                assert inp[row[e]] in ix[ip[inp[col[e]]]:ip[inp[col[e]] + 1]]
        assert got[f"{key}:feat"].tobytes() == oracle.extract(d["feat"], inp).astype(np.float32).tobytes()


def _oracle_batches(d, worker_id, num_worker, batch_size, num_epoch, fanouts, seed, arch6, sample_type="khop3",
                    nstates=None, states=None, **kw):
    """Replays shuffler + sampler + extract on the CPU exactly as the engine is specified to."""
    train = d["train"]
    padded = oracle.aligned_pad(train, num_worker)
    n_local = padded.size // num_worker
    n_local_step = (n_local + batch_size - 1) // batch_size
    n_global = n_local_step * num_worker
    max_seeds = int(batch_size * 1.25) + 1
    nstates = max(oracle.predict_num_nodes(max_seeds, fanouts, len(fanouts) - 1),
                  (oracle.predict_num_nodes(max_seeds, fanouts, len(fanouts) - 1) + 127) // 128 * 8,
                  (oracle.predict_num_nodes(max_seeds, fanouts, len(fanouts) - 1) + 1023) // 1024 * 256)
    if sample_type in ("weighted_khop", "khop1", "weighted_khop_prefix"):
        nstates = min(oracle.predict_num_nodes(max_seeds, fanouts, len(fanouts)), 512 * 1024)
    if sample_type == "random_walk":
        nstates = (oracle.predict_num_nodes(max_seeds, fanouts, len(fanouts) - 1) + 63) // 64 * 256
    if states is None:
        states = oracle.random_states(nstates, seed + 1000003 * worker_id)
    data = padded.copy()
    out = {}
    for ep in range(num_epoch):
        data = oracle.shuffle_minstd0(data, ep if arch6 else seed + ep)
        local = data[worker_id * n_local:(worker_id + 1) * n_local]
        for st in range(n_local_step):
            off = st * batch_size
            size = min(batch_size, n_local - off)
            if arch6 and ep == 0 and st == 0:
                size = min(int(size * 1.25), n_local - off)
            seeds = local[off:off + size]
            code = {"khop3": oracle.KHOP3, "khop0": oracle.KHOP0, "khop2": oracle.KHOP2, "khop1": oracle.KHOP1,
                    "weighted_khop_prefix": oracle.WEIGHTED_KHOP_PREFIX,
                    "weighted_khop_hash_dedup": oracle.WEIGHTED_KHOP_HASH_DEDUP, "weighted_khop": oracle.WEIGHTED_KHOP,
                    "random_walk": oracle.RANDOM_WALK}[sample_type]
            res = oracle.do_sample(code, d["ip"], d["ix"], seeds, fanouts, states, **kw)
            key = ep * n_global + worker_id * n_local_step + st
            out[key] = dict(res=res, seeds=seeds, feat=oracle.extract(d["feat"], res["input_nodes"]),
                            label=d["label"][seeds])
    return out


def _check(npz, want, num_layers):
    keys = sorted({int(k.split(":")[0]) for k in npz.files})
    assert keys == sorted(want.keys())
    for key in keys:
        w = want[key]
        np.testing.assert_array_equal(npz[f"{key}:output_nodes"].view(np.uint32), w["seeds"])
        np.testing.assert_array_equal(npz[f"{key}:input_nodes"].view(np.uint32), w["res"]["input_nodes"])
        for i in range(num_layers):
            np.testing.assert_array_equal(npz[f"{key}:row{i}"].view(np.uint32), w["res"]["layers"][i]["row"])
            np.testing.assert_array_equal(npz[f"{key}:col{i}"].view(np.uint32), w["res"]["layers"][i]["col"])
            assert int(npz[f"{key}:num_src{i}"]) == w["res"]["layers"][i]["num_src"]
            assert int(npz[f"{key}:num_dst{i}"]) == w["res"]["layers"][i]["num_dst"]
            if w["res"]["layers"][i]["data"] is not None:
                np.testing.assert_array_equal(npz[f"{key}:data{i}"].view(np.uint32), w["res"]["layers"][i]["data"])
        assert npz[f"{key}:feat"].tobytes() == w["feat"].astype(np.float32).tobytes()
        np.testing.assert_array_equal(npz[f"{key}:label"], w["label"])
        assert float(npz[f"{key}:num_sample"]) == sum(l["row"].size for l in w["res"]["layers"])


@pytest.mark.gpu
@pytest.mark.parametrize("pipelined,sample_type,table", [(0, "khop3", "direct"), (1, "khop3", "hashed"), (0, "khop0", "direct"),
                                                         (1, "khop2", "direct"), (0, "khop1", "direct")])
def test_arch1_end_to_end(tmp_path, pipelined, sample_type, table):
    d = make_dataset(tmp_path / "ds")
    prefix = str(tmp_path / "out")
    # sample_once() enqueues one batch ahead by default; the hashed-table case runs without that
    r = subprocess.run([sys.executable, DRIVER, d["path"], prefix, "arch1", "1", f"pipelined={pipelined}",
                        f"sample_type={sample_type}", f"hash_table={table}", "seed=99", "batch_size=64", "fanout=5 4",
                        f"lookahead={0 if table == 'hashed' else 1}"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    want = _oracle_batches(d, 0, 1, 64, 2, [5, 4], 99, arch6=False, sample_type=sample_type)
    _check(np.load(prefix + ".w0.npz"), want, 2)


@pytest.mark.gpu
@pytest.mark.parametrize("streams,pipelined", [(1, 0), (2, 0), (2, 1)])
def test_arch1_one_and_two_extract_streams_give_the_oracles_batches(tmp_path, streams, pipelined):
    """Config key `extract_streams` (an extension; default 2): consecutive lean batches' gathers alternate between two
    extract streams and may overlap; with three batches enqueued ahead (lookahead 2) two gathers and a sampler are in
    flight at once.  Every batch still equals the oracle's, whoever finishes first."""
    d = make_dataset(tmp_path / "ds")
    prefix = str(tmp_path / "out")
    r = subprocess.run([sys.executable, DRIVER, d["path"], prefix, "arch1", "1", f"pipelined={pipelined}", "sample_type=khop3",
                        "seed=7", "batch_size=64", "fanout=5 4", "lookahead=2", f"extract_streams={streams}", "num_epoch=3"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    want = _oracle_batches(d, 0, 1, 64, 3, [5, 4], 7, arch6=False)
    _check(np.load(prefix + ".w0.npz"), want, 2)


@pytest.mark.gpu
def test_arch1_weighted_and_random_walk(tmp_path):
    """PinSAGE (random walk, train_pinsage.py defaults) and weighted sampling through the engine."""
    d = make_dataset(tmp_path / "ds")
    from xgnn_amd import datagen
    # VALID tables, built from per-edge weights the way the reference's weight tools build them
    # (create_alias_table.cc:105-170, create_prob_prefix_table.cc:94-123) and written by write_dataset
    g = dict(indptr=d["ip"], indices=d["ix"], train_set=d["train"], meta=dict(feat_dim=d["feat"].shape[1], num_class=13))
    weights = datagen.edge_weights(g, "default", seed=3)
    datagen.write_dataset(d["path"], g, feat=d["feat"], label=d["label"], weights=weights)
    prob = np.fromfile(os.path.join(d["path"], "prob_table.bin"), np.float32)
    alias = np.fromfile(os.path.join(d["path"], "alias_table.bin"), np.uint32)
    pre = np.fromfile(os.path.join(d["path"], "prob_prefix_table.bin"), np.float32)
    po, ao = oracle.create_alias_table(d["ip"], d["ix"], weights)
    assert prob.tobytes() == po.tobytes() and np.array_equal(alias, ao)
    for stype, fan, kw in [("weighted_khop", [5, 4], dict(prob=prob, alias=alias)),
                           ("weighted_khop_prefix", [5, 4], dict(prob=pre)),
                           ("weighted_khop_hash_dedup", [5, 4], dict(prob=prob, alias=alias)),
                           ("random_walk", [5, 5, 5], dict(walk_length=3, restart_prob=0.5, num_walk=4))]:
        prefix = str(tmp_path / f"out_{stype}")
        r = subprocess.run([sys.executable, DRIVER, d["path"], prefix, "arch1", "1", f"sample_type={stype}", "seed=21",
                            "num_epoch=1", "fanout=" + " ".join(map(str, fan))],
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        want = _oracle_batches(d, 0, 1, 64, 1, fan, 21, arch6=False, sample_type=stype, **kw)
        _check(np.load(prefix + ".w0.npz"), want, len(fan))


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [np.float16, np.uint8])
def test_arch1_feature_dtypes(tmp_path, dtype):
    """get_graph_feat casts non-f32 tables to float (adapter.py:118-122)."""
    d = make_dataset(tmp_path / "ds", dim=7, dtype=dtype)
    prefix = str(tmp_path / "out")
    r = subprocess.run([sys.executable, DRIVER, d["path"], prefix, "arch1", "1", "seed=5", "num_epoch=1"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    want = _oracle_batches(d, 0, 1, 64, 1, [5, 4], 5, arch6=False)
    _check(np.load(prefix + ".w0.npz"), want, 2)


@pytest.mark.gpu
@pytest.mark.parametrize("opts", [
    dict(cache_percentage="0.4", part_cache="True", gpu_extract="True", use_dist_graph="1.0"),
    dict(cache_percentage="0.25", gpu_extract="True", use_dist_graph="0.5"),
    dict(cache_percentage="0", gpu_extract="True"),
    # `gpu_extract` off = the reference's SGNN mode: miss rows gathered by the host threads into pinned memory and
    # copied down asynchronously (dist_loops.cc:1076-1207), hits from the (partitioned) cache
    dict(cache_percentage="0.3"),
    dict(cache_percentage="0.3", part_cache="True", use_dist_graph="1.0"),
    # the same path as the reference runs it -- one phase at a time, each behind its own wait (first epoch), then chunked
    dict(cache_percentage="0.3", staged_serial_epochs="1"),
    # cache 0 without gpu_extract (DoIdCopy + DoCPUFeatureExtract + DoFeatureCopy, dist_loops_arch6.cc:111-133):
    # every row gathered by the host team, chunks copied straight into the batch
    dict(cache_percentage="0"),
    dict(cache_percentage="0", staged_serial_epochs="2"),
    # hybrid store: the hotter half of the cached slots on every GPU, the rest sharded (replica + shards + host rows
    # behind one gather, ggms_extract_tiered)
    dict(cache_percentage="0.4", part_cache="True", gpu_extract="True", use_dist_graph="1.0", replicate_percentage="0.5"),
    dict(cache_percentage="1.0", part_cache="True", gpu_extract="True", replicate_percentage="0.25"),
])
def test_arch6_two_workers_one_gpu(tmp_path, opts):
    """XGNN mode: topology shards + partitioned feature cache shared through hipIpc, two forked workers
    (both mapped onto the single GPU of the test box by SAMGRAPH_FORCE_DEVICE)."""
    d = make_dataset(tmp_path / "ds")
    prefix = str(tmp_path / "out")
    # the host-staged path moves its miss rows in chunks: small ones here, so that a batch takes several
    env = dict(os.environ, SAMGRAPH_FORCE_DEVICE="0", HSA_ENABLE_IPC_MODE_LEGACY="0", SAMGRAPH_STAGED_CHUNK_ROWS="100")
    r = subprocess.run([sys.executable, DRIVER, d["path"], prefix, "arch6", "2", "seed=7", "batch_size=64",
                        "fanout=5 4"] + [f"{k}={v}" for k, v in opts.items()],
                       capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    for w in range(2):
        want = _oracle_batches(d, w, 2, 64, 2, [5, 4], 7, arch6=True)
        npz = np.load(f"{prefix}.w{w}.npz")
        _check(npz, want, 2)
        if float(opts["cache_percentage"]) == 0 and "gpu_extract" not in opts:  # host-staged, no cache: every row is a miss
            for key, wv in want.items():
                assert float(npz[f"{key}:miss_bytes"]) == wv["res"]["input_nodes"].size * d["feat"].shape[1] * 4
        if float(opts["cache_percentage"]) > 0:
            # misses are exactly the input nodes outside the cached prefix of the rank list
            from xgnn_amd import datagen
            rank = datagen.degree_rank(d["ip"])
            ncache = int(d["ip"].size - 1) * float(opts["cache_percentage"])
            cached = np.zeros(d["ip"].size - 1, bool)
            cached[rank[: int(ncache)]] = True
            for key, wv in want.items():
                nmiss = int((~cached[wv["res"]["input_nodes"]]).sum())
                assert float(npz[f"{key}:miss_bytes"]) == nmiss * d["feat"].shape[1] * 4


@pytest.mark.gpu
def test_arch6_padding_copies_in_one_batch_take_the_general_path(tmp_path):
    """the engine promises distinct seeds to the sampler batch by batch (train set checked once, the padding
    copies of the aligned epoch located per Reshuffle).  Here the promise must be WITHDRAWN: 100 train nodes over three
    workers are padded with two copies (dist_shuffler_aligned.cc:52-54), a worker's whole slice of 34 is one batch, so
    in some epochs both copies of a padding node are seeds of one batch -- those batches must come out as the oracle's
    (duplicated seeds deduplicated by FillWithDupRevised), the others through the distinct-seed path."""
    d = make_dataset(tmp_path / "ds", num_train=100)
    prefix = str(tmp_path / "out")
    W, epochs = 3, 8
    env = dict(os.environ, SAMGRAPH_FORCE_DEVICE="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, DRIVER, d["path"], prefix, "arch6", str(W), "seed=7", "batch_size=64", "fanout=5 4",
                        "cache_percentage=0.4", "part_cache=True", "gpu_extract=True", f"num_epoch={epochs}"],
                       capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    both = 0
    for w in range(W):
        want = _oracle_batches(d, w, W, 64, epochs, [5, 4], 7, arch6=True)
        _check(np.load(f"{prefix}.w{w}.npz"), want, 2)
        both += sum(np.unique(v["seeds"]).size != v["seeds"].size for v in want.values())
    assert both >= 1  # the case this test exists for did occur (the shuffle is seeded by the epoch number)


@pytest.mark.gpu
def test_arch6_five_workers_one_gpu(tmp_path):
    """Rehearsal of the 8-GPU deployment as far as a one-GPU box allows (its process guard admits six processes on
    the card, and the test runner is one of them): five forked workers, five-way topology shards + partitioned
    feature cache + hot-row replica, every worker opening four peers through hipIpc, the deadline-carrying worker
    barrier -- each worker against the oracle replay of ITS slice of the epoch (DistAlignedShuffler)."""
    d = make_dataset(tmp_path / "ds")
    prefix = str(tmp_path / "out")
    W = 5
    env = dict(os.environ, SAMGRAPH_FORCE_DEVICE="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, DRIVER, d["path"], prefix, "arch6", str(W), "seed=7", "batch_size=32", "fanout=5 4",
                        "cache_percentage=0.4", "part_cache=True", "gpu_extract=True", "use_dist_graph=1.0",
                        "replicate_percentage=0.25", "num_epoch=1"],
                       capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    for w in range(W):
        _check(np.load(f"{prefix}.w{w}.npz"), _oracle_batches(d, w, W, 32, 1, [5, 4], 7, arch6=True), 2)


@pytest.mark.gpu
def test_arch6_worker_that_never_arrives_ends_the_run(tmp_path):
    """A worker that dies before publishing its shards: the others must not wait forever at the worker barrier
    (pthread_barrier_wait would) -- they abort with a message once SAMGRAPH_IPC_TIMEOUT_S has passed."""
    import time
    d = make_dataset(tmp_path / "ds")
    env = dict(os.environ, SAMGRAPH_FORCE_DEVICE="0", HSA_ENABLE_IPC_MODE_LEGACY="0", SAMGRAPH_IPC_TIMEOUT_S="4")
    t0 = time.time()
    r = subprocess.run([sys.executable, DRIVER, d["path"], str(tmp_path / "out"), "arch6", "2", "seed=7", "batch_size=64",
                        "fanout=5 4", "cache_percentage=0.4", "part_cache=True", "gpu_extract=True", "use_dist_graph=1.0",
                        "die_worker=1"], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode != 0 and time.time() - t0 < 120
    assert "worker barrier" in r.stderr and "1 of 2 workers arrived" in r.stderr, r.stderr[-2000:]


@pytest.mark.gpu
def test_arch6_staged_extract_uses_the_configured_host_team(tmp_path):
    """`gpu_extract` off (the reference's SGNN mode): the miss rows are gathered by omp_thread_num host threads
    (cuda_cache_manager_host.cc:268-300 `omp parallel for num_threads(omp_thread_num)`), under arch6 too."""
    d = make_dataset(tmp_path / "ds")
    prefix = str(tmp_path / "out")
    env = dict(os.environ, SAMGRAPH_FORCE_DEVICE="0", HSA_ENABLE_IPC_MODE_LEGACY="0", SAMGRAPH_LOG_LEVEL="info")
    r = subprocess.run([sys.executable, DRIVER, d["path"], prefix, "arch6", "1", "seed=7", "batch_size=64", "fanout=5 4",
                        "cache_percentage=0.3", "omp_thread_num=5", "num_epoch=1"],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    assert "host team of 5 threads" in r.stderr
    _check(np.load(f"{prefix}.w0.npz"), _oracle_batches(d, 0, 1, 64, 1, [5, 4], 7, arch6=True), 2)


@pytest.mark.gpu
def test_cpp_driver_over_the_c_abi(tmp_path):
    """A pure C++ caller (tools/samgraph_no_train.cc, the role of samgraph/main.cc) drives the same ABI."""
    exe = os.path.join(ROOT, "build", "samgraph_no_train")
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "xgnn_amd", "csrc"), "driver"])
    d = make_dataset(tmp_path / "ds")
    r = subprocess.run([exe, "--dataset-path", d["path"], "--batch-size", "64", "--num-epoch", "2", "--fanout", "5 4",
                        "--seed", "3"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("[epoch")]
    assert len(lines) == 2 and "8 steps" in lines[0]
    # same seed => same number of sampled edges as the oracle replay
    want = _oracle_batches(d, 0, 1, 64, 2, [5, 4], 3, arch6=False)
    for ep, line in enumerate(lines):
        got_edges = int(line.split("| ")[1].split(" edges")[0])
        edges = sum(sum(l["row"].size for l in w["res"]["layers"]) for k, w in want.items() if k // 8 == ep)
        assert got_edges == edges


@pytest.mark.gpu
def test_example_training_loop(tmp_path):
    """tools/example_train_sage.py: the reference's loop (config / init / sample_once / get_next_batch / get_graph_*)
    feeding a plain-PyTorch GraphSAGE on the GPU -- the engine's zero-copy tensors are usable by torch as they are."""
    d = make_dataset(tmp_path / "ds", num_node=4000, dim=16, num_train=1500)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "example_train_sage.py"), d["path"], "--epochs", "3",
                        "--batch-size", "256", "--fanout", "5", "4"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("epoch ")]
    assert len(lines) == 3
    losses = [float(l.split("loss ")[1].split(",")[0]) for l in lines]
    assert all(np.isfinite(losses)) and losses[-1] < losses[0]


@pytest.mark.gpu
def test_arch6_presample_cache_policy(tmp_path):
    """cache_policy = pre_sample (dist/pre_sampler.cc:39-139): worker 0 samples `presample_epoch` epochs of the
    whole train set, ranks nodes by (visit count << 32 | id) descending, and that ranking fills the cache."""
    d = make_dataset(tmp_path / "ds")
    prefix = str(tmp_path / "out")
    env = dict(os.environ, SAMGRAPH_FORCE_DEVICE="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    seed, bs, fan, ratio = 17, 64, [5, 4], 0.3
    r = subprocess.run([sys.executable, DRIVER, d["path"], prefix, "arch6", "2", f"seed={seed}", f"batch_size={bs}",
                        "fanout=5 4", "cache_policy=pre_sample", "presample_epoch=2", f"cache_percentage={ratio}",
                        "part_cache=True", "gpu_extract=True"], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    # oracle replay of the presample on worker 0's RNG states
    max_seeds = int(bs * 1.25) + 1
    nst = oracle.predict_num_nodes(max_seeds, fan, len(fan) - 1)
    st0 = oracle.random_states(nst, seed)
    N = d["ip"].size - 1
    freq = np.zeros(N, np.uint64)
    train = d["train"].copy()
    for e in range(2):
        train = oracle.shuffle_minstd0(train, seed + 0x5A5A5A + e)
        for off in range(0, train.size, bs):
            res = oracle.do_sample(oracle.KHOP3, d["ip"], d["ix"], train[off:off + bs], fan, st0)
            freq[res["input_nodes"]] += 1
    keys = (freq << np.uint64(32)) | np.arange(N, dtype=np.uint64)
    rank = (np.sort(keys)[::-1] & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    cached = np.zeros(N, bool)
    cached[rank[: int(N * ratio)]] = True
    for w in range(2):
        want = _oracle_batches(d, w, 2, bs, 2, fan, seed, arch6=True, states=st0 if w == 0 else None)
        npz = np.load(f"{prefix}.w{w}.npz")
        _check(npz, want, 2)
        for key, wv in want.items():
            nmiss = int((~cached[wv["res"]["input_nodes"]]).sum())
            assert float(npz[f"{key}:miss_bytes"]) == nmiss * d["feat"].shape[1] * 4
