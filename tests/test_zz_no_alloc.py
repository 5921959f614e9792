"""tl_attention_noalloc (the entry point that allocates nothing; include/tlhip.h) against tl_attention through the
C-ABI: same kernels over a static device array instead of the hipMalloc workspace, so the outputs must be BIT-IDENTICAL
(and both are within the attention bar of the oracle). Two calls in a row, the second with a smaller input: the array
keeps the first call's partials, none of which the second call may read. Collected last on purpose (file name)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def E(pkg):
    from tinyllama_cpp_amd import engine
    engine.init(0)
    return engine


def _run(E, lib, check, fn, q, kc, vc, nh, nkv, hd, npos, max_seq):
    qb = E.DeviceBuffer.from_numpy(q)
    kb = E.DeviceBuffer.from_numpy(kc)
    vb = E.DeviceBuffer.from_numpy(vc)
    ob = E.DeviceBuffer(4 * nh * hd)
    check(lib().tl_memset(ob.ptr, 0xff, ob.nbytes, None))  # poison: every output element must be written
    scale = float(np.float32(1.0) / np.sqrt(np.float32(hd)))
    check(fn(qb.ptr, kb.ptr, vb.ptr, ob.ptr, nh, npos, hd, scale, max_seq, nkv, None), fn.__name__)
    check(lib().tl_device_sync())
    return ob.to_numpy(np.float32, (nh * hd,))


def test_noalloc_attention_matches_tl_attention_twice(E, oracle):
    from tinyllama_cpp_amd.capi import lib, check
    # first a long sequence on many heads (fills 32 splits of every head), then a short one on fewer heads
    for nh, nkv, hd, npos in ((64, 8, 128, 1900), (32, 8, 128, 70), (32, 4, 64, 3)):
        rng = np.random.default_rng(npos)
        max_seq = npos + 3
        kc = np.zeros((max_seq, nkv * hd), np.float32)
        vc = np.zeros((max_seq, nkv * hd), np.float32)
        kc[:npos] = rng.standard_normal((npos, nkv * hd))
        vc[:npos] = rng.standard_normal((npos, nkv * hd))
        q = rng.standard_normal(nh * hd).astype(np.float32)
        got = _run(E, lib, check, lib().tl_attention_noalloc, q, kc, vc, nh, nkv, hd, npos, max_seq)
        ref = _run(E, lib, check, lib().tl_attention, q, kc, vc, nh, nkv, hd, npos, max_seq)
        assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), np.abs(got - ref).max()
        want = oracle.attention_decode(q, kc, vc, nh, nkv, hd, npos - 1)
        assert np.allclose(got, want, rtol=0, atol=2e-5), np.abs(got - want).max()


def test_noalloc_attention_rejects_what_does_not_fit(E):
    from tinyllama_cpp_amd.capi import lib
    b = E.DeviceBuffer(4 * 128 * 128 * 4)
    # 128 heads x 128: twice the static workspace; 8 query heads per kv head keeps it on the split path
    rc = lib().tl_attention_noalloc(b.ptr, b.ptr, b.ptr, b.ptr, 128, 1, 128, 1.0, 4, 16, None)
    assert rc != 0 and b"static workspace" in lib().tl_last_error()


def test_noalloc_attention_allocates_nothing_and_runs_under_capture(E, oracle):
    """What the entry point exists for: after a warm-up call (which sets the kernel attributes of its instantiations) a
    call leaves the device's free memory unchanged, and it can be recorded by stream capture (an allocation inside a
    capturing stream is an error) and replayed as a graph, with the same bits as the direct call."""
    import ctypes as C
    from tinyllama_cpp_amd.capi import lib, check
    hip = C.CDLL("libamdhip64.so")
    nh, nkv, hd, npos = 32, 8, 128, 700
    rng = np.random.default_rng(5)
    max_seq = npos + 1
    kc = rng.standard_normal((max_seq, nkv * hd)).astype(np.float32)
    vc = rng.standard_normal((max_seq, nkv * hd)).astype(np.float32)
    q = rng.standard_normal(nh * hd).astype(np.float32)
    qb, kb, vb = (E.DeviceBuffer.from_numpy(a) for a in (q, kc, vc))
    ob, og = E.DeviceBuffer(4 * nh * hd), E.DeviceBuffer(4 * nh * hd)
    scale = float(np.float32(1.0) / np.sqrt(np.float32(hd)))
    fn = lib().tl_attention_noalloc
    check(fn(qb.ptr, kb.ptr, vb.ptr, ob.ptr, nh, npos, hd, scale, max_seq, nkv, None), "warm-up")
    check(lib().tl_device_sync())

    def free_bytes():
        f, t = C.c_size_t(), C.c_size_t()
        assert hip.hipMemGetInfo(C.byref(f), C.byref(t)) == 0
        return f.value

    before = free_bytes()
    for _ in range(3):
        check(fn(qb.ptr, kb.ptr, vb.ptr, ob.ptr, nh, npos, hd, scale, max_seq, nkv, None), "direct call")
    check(lib().tl_device_sync())
    assert free_bytes() == before
    direct = ob.to_numpy(np.float32, (nh * hd,))
    # capture one call on a stream of its own, replay the graph twice into a poisoned buffer
    s = C.c_void_p()
    check(lib().tl_stream_create(C.byref(s)), "stream")
    graph, gexec = C.c_void_p(), C.c_void_p()
    assert hip.hipStreamBeginCapture(s, 0) == 0  # hipStreamCaptureModeGlobal: unsafe calls (hipMalloc) would fail
    rc = fn(qb.ptr, kb.ptr, vb.ptr, og.ptr, nh, npos, hd, scale, max_seq, nkv, s)
    assert hip.hipStreamEndCapture(s, C.byref(graph)) == 0
    assert rc == 0, lib().tl_last_error()
    assert hip.hipGraphInstantiate(C.byref(gexec), graph, None, None, C.c_size_t(0)) == 0
    for _ in range(2):
        check(lib().tl_memset(og.ptr, 0xff, og.nbytes, s))
        assert hip.hipGraphLaunch(gexec, s) == 0
        check(lib().tl_stream_sync(s))
        got = og.to_numpy(np.float32, (nh * hd,))
        assert np.array_equal(got.view(np.uint32), direct.view(np.uint32))
    assert free_bytes() <= before  # (the graph objects themselves may take device memory; the calls did not)
    hip.hipGraphExecDestroy(gexec)
    hip.hipGraphDestroy(graph)
    check(lib().tl_stream_destroy(s))
    want = oracle.attention_decode(q, kc, vc, nh, nkv, hd, npos - 1)
    assert np.allclose(direct, want, rtol=0, atol=2e-5)
