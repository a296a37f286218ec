"""ggms_launch_timer_t (include/ggms.h): a row gather's own start / end timestamps on its dispatch packet.

The timed gather must produce the oracle's bytes like any other, its duration must agree with an event pair around
the same launch, its end event must order another stream behind it, and a timer rides exactly one launch."""
import numpy as np
import pytest

import oracle

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU (run with -m gpu on an MI355X box)")
    from xgnn_amd import ops as o
    return o


def _table(n, dim, dev):
    feat = (np.arange(n * dim, dtype=np.int64) & 0xFFFF).astype(np.float32).reshape(n, dim)
    return feat, torch.from_numpy(feat).to(dev)


def test_timed_gather_is_the_oracles_gather_and_its_time_agrees_with_an_event_pair(ops):
    dev = torch.device("cuda", 0)
    n, dim, rows = 1 << 20, 128, 1 << 19  # 256 MB of rows gathered: a few hundred microseconds
    feat, t_feat = _table(n, dim, dev)
    idx = np.random.RandomState(5).randint(0, n, rows).astype(np.uint32)
    t_idx = torch.from_numpy(idx.view(np.int32)).to(dev)
    out = torch.empty((rows, dim), dtype=torch.float32, device=dev)
    tm = ops.LaunchTimer()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(2):  # second pass: warm
        out.zero_()
        torch.cuda.synchronize()
        e0.record()
        tm.arm()
        ops.extract(t_feat, t_idx, out=out)
        e1.record()
        us = tm.elapsed_us()  # blocks until the launch is done
        torch.cuda.synchronize()
    assert out.cpu().numpy().tobytes() == oracle.extract(feat, idx).tobytes()
    pair_us = e0.elapsed_time(e1) * 1e3
    assert us > 20.0, us
    # the pair also holds its own two marker packets: the kernel's own time is a little less, never much more
    assert 0.5 * pair_us <= us <= 1.05 * pair_us + 5.0, (us, pair_us)
    tm.close()


def test_a_timer_rides_one_launch_and_orders_another_stream_behind_it(ops):
    dev = torch.device("cuda", 0)
    n, dim, rows = 1 << 18, 100, 1 << 17
    feat, t_feat = _table(n, dim, dev)
    idx = np.random.RandomState(6).randint(0, n, rows).astype(np.uint32)
    t_idx = torch.from_numpy(idx.view(np.int32)).to(dev)
    out = torch.zeros((rows, dim), dtype=torch.float32, device=dev)
    copy = torch.zeros_like(out)
    tm = ops.LaunchTimer()
    from xgnn_amd._lib import GgmsError
    with pytest.raises(GgmsError):  # never rode a launch
        tm.elapsed_us()
    tm.wait()  # a no-op before any launch
    s1, s2 = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
    torch.cuda.synchronize()
    tm.arm()
    with torch.cuda.stream(s1):
        ops.extract(t_feat, t_idx[:0], out=out[:0])  # no rows, no launch: the timer stays armed
        ops.extract(t_feat, t_idx, out=out)          # this one carries it
    with torch.cuda.stream(s2):
        tm.wait(s2)                                  # "the rows are out"
        copy.copy_(out)
    first = tm.elapsed_us()
    with torch.cuda.stream(s1):
        ops.extract(t_feat, t_idx[:1024], out=out[:1024])  # not timed: the timer was consumed
    torch.cuda.synchronize()
    assert tm.elapsed_us() == first
    assert copy.cpu().numpy().tobytes() == oracle.extract(feat, idx).tobytes()
    # re-armed: the next launch's time replaces the first
    tm.arm()
    ops.extract(t_feat, t_idx[:4096], out=out[:4096])
    again = tm.elapsed_us()
    assert 0.0 < again < first
    tm.close()


def test_span_of_two_timed_launches_on_two_streams(ops):
    """ggms_launch_timer_span_us: first launch's start -> last launch's end.  Two gathers issued on two streams may
    overlap: the span is of the order of one launch at least and never longer than the two back to back plus slack."""
    dev = torch.device("cuda", 0)
    n, dim, rows = 1 << 20, 128, 1 << 19
    feat, t_feat = _table(n, dim, dev)
    idx = np.random.RandomState(8).randint(0, n, rows).astype(np.uint32)
    t_idx = torch.from_numpy(idx.view(np.int32)).to(dev)
    outs = [torch.empty((rows, dim), dtype=torch.float32, device=dev) for _ in range(2)]
    tms = [ops.LaunchTimer(), ops.LaunchTimer()]
    streams = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]
    from xgnn_amd._lib import GgmsError
    with pytest.raises(GgmsError):
        tms[0].span_us(tms[1])  # neither rode a launch yet
    for _ in range(2):  # second pass: warm
        torch.cuda.synchronize()
        for t, s, o in zip(tms, streams, outs):
            with torch.cuda.stream(s):
                t.arm()
                ops.extract(t_feat, t_idx, out=o)
        a, b = tms[0].elapsed_us(), tms[1].elapsed_us()
        span = tms[0].span_us(tms[1])
    torch.cuda.synchronize()
    assert span >= 0.5 * min(a, b) and span <= a + b + 200.0, (a, b, span)
    want = oracle.extract(feat, idx).tobytes()
    assert outs[0].cpu().numpy().tobytes() == want and outs[1].cpu().numpy().tobytes() == want
    for t in tms:
        t.close()
