"""The entry point that allocates nothing (mfnet_colsum_static): called through the C ABI, checked against mfnet_colsum (the
existing entry point with a caller workspace) bit for bit, on two calls in a row, the second with a smaller input - the
static scratch keeps the first call's slabs and the tickets must have been cleared."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _both(x, G, P, C, want_sq):
    from ssrs_amd import _lib
    from ssrs_amd._lib import BF16, F32
    dt = BF16 if x.dtype == torch.bfloat16 else F32
    st = torch.cuda.current_stream().cuda_stream
    ref = torch.zeros((2, G, C), dtype=torch.float32, device="cuda")
    nws = int(_lib.lib().mfnet_colreduce_workspace_floats(G, P, C))
    ws = torch.empty(nws, dtype=torch.float32, device="cuda")
    tk = torch.zeros(G, dtype=torch.int32, device="cuda")
    _lib.call("mfnet_colsum", x.data_ptr(), dt, ref[0].data_ptr(), ref[1].data_ptr() if want_sq else None, G, P, C, ws.data_ptr(), tk.data_ptr(), st)
    got = torch.zeros((2, G, C), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    # the DEVICE's free bytes (hipMemGetInfo), not torch's allocator statistics: a hipMalloc / hipMallocAsync inside the library
    # would not show in torch.cuda.memory_allocated()
    before, free_before = torch.cuda.memory_allocated(), torch.cuda.mem_get_info()[0]
    _lib.call("mfnet_colsum_static", x.data_ptr(), dt, got[0].data_ptr(), got[1].data_ptr() if want_sq else None, G, P, C, st)
    torch.cuda.synchronize()
    assert torch.cuda.memory_allocated() == before and torch.cuda.mem_get_info()[0] == free_before
    return got, ref


def test_colsum_static_matches_colsum_on_two_calls_in_a_row():
    from ssrs_amd import _lib
    g = torch.Generator(device="cuda").manual_seed(5)
    # first call: a large map (16 groups x 4096 rows x 256 channels, bf16: the SE pool of the 64x64 FPN level)
    x1 = torch.randn((16 * 4096, 256), generator=g, device="cuda").to(torch.bfloat16)
    got, ref = _both(x1, 16, 4096, 256, True)
    assert torch.equal(got, ref) and ref[0].abs().sum() > 0
    # second call right after it: a smaller fp32 map with fewer groups, splits and channels (BatchNorm statistics)
    x2 = torch.randn((3 * 500, 64), generator=g, device="cuda")
    got, ref = _both(x2, 3, 500, 64, True)
    assert torch.equal(got, ref)
    assert torch.allclose(ref[0], x2.view(3, 500, 64).sum(1), rtol=1e-4, atol=1e-3)
    # and again the large one, sums only
    got, ref = _both(x1, 16, 4096, 256, False)
    assert torch.equal(got, ref) and float(got[1].abs().sum()) == 0.0
    # sizes beyond the static scratch are refused with a clear message, nothing is launched
    big = torch.empty((2048 * 8, 1024), dtype=torch.bfloat16, device="cuda")
    out = torch.zeros((2048, 1024), dtype=torch.float32, device="cuda")
    with pytest.raises(_lib.MfnetError, match="static scratch"):
        _lib.call("mfnet_colsum_static", big.data_ptr(), 1, out.data_ptr(), None, 2048, 8, 1024, torch.cuda.current_stream().cuda_stream)


def test_colsum_static_under_graph_capture():
    """Its stated callers cannot allocate - code under HIP graph capture first of all: the entry point is captured (after one
    plain call, which takes the scratch's address once per device), replayed twice on changing input, and checked against
    mfnet_colsum bit for bit."""
    from ssrs_amd import _lib
    G, P, C = 4, 1000, 128
    g = torch.Generator(device="cuda").manual_seed(9)
    x = torch.randn((G * P, C), generator=g, device="cuda")
    got = torch.zeros((2, G, C), dtype=torch.float32, device="cuda")
    first, _ = _both(x, G, P, C, True)                                   # plain call: per-device set-up done outside the capture
    side = torch.cuda.Stream()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side, capture_error_mode="thread_local"):
        _lib.call("mfnet_colsum_static", x.data_ptr(), _lib.F32, got[0].data_ptr(), got[1].data_ptr(), G, P, C, torch.cuda.current_stream().cuda_stream)
    for k in range(2):
        x.copy_(torch.randn((G * P, C), generator=g, device="cuda"))
        got.zero_()
        torch.cuda.synchronize()
        graph.replay()
        torch.cuda.synchronize()
        ref = torch.zeros((2, G, C), dtype=torch.float32, device="cuda")
        nws = int(_lib.lib().mfnet_colreduce_workspace_floats(G, P, C))
        ws = torch.empty(nws, dtype=torch.float32, device="cuda")
        tk = torch.zeros(G, dtype=torch.int32, device="cuda")
        _lib.call("mfnet_colsum", x.data_ptr(), _lib.F32, ref[0].data_ptr(), ref[1].data_ptr(), G, P, C, ws.data_ptr(), tk.data_ptr(),
                  torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        assert torch.equal(got, ref) and not torch.equal(ref, first), k
