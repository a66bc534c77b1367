"""Asynchronous submissions (fbr_gram_submit / fbr_tsqr_submit / fbr_wait): ownership and ordering rules of the boundary.

  * temporaries the binding makes on the way (contiguous copies of strided tensors, reshapes) stay alive until wait();
  * switching the stream (fbr_model_set_stream) or calling a blocking entry point first completes what is in flight;
  * a submission that fails issues no ticket and leaves nothing in flight; the engine keeps working;
  * fbr_tsqr_submit takes device-resident data only, with or without row weights, all columns or a subset."""
import numpy as np
import pytest

from common import load_topo, random_states

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def setup():
    import torch

    from flobaroid_amd._lib import Engine

    topo = load_topo("walkman_apriori")
    S = 60000
    rng = np.random.default_rng(21)
    st_np = random_states(topo, S, rng, 1, use_limits=True)
    dev = torch.device("cuda", 0)
    st = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in st_np.items()}
    eng = Engine(topo, floating=True)
    eng.use_torch_stream()
    rhs = torch.randn((S * eng.rows, 2), dtype=torch.float64, device=dev)
    yield topo, eng, st, rhs, dev, S
    eng.close()


def test_strided_inputs_are_kept_alive_until_wait(setup):
    """Non-contiguous CUDA tensors are copied by the binding; those copies must not go back to torch's allocator while the GPU reads them."""
    import torch

    topo, eng, st, rhs, dev, S = setup
    G0 = eng.gram(st, rhs=rhs).clone()
    wide = {k: torch.cat([v, v], dim=1) for k, v in st.items()}
    strided = {k: wide[k][:, : st[k].shape[1]] for k in st}          # views with a doubled row stride
    assert not strided["q"].is_contiguous()
    rhs3 = rhs.reshape(S, eng.rows, 2)                                  # reshaped on the way in
    out = torch.zeros_like(G0)
    t = eng.gram_submit(strided, out, rhs=rhs3)
    junk = [torch.full((S, st["q"].shape[1]), float("nan"), dtype=torch.float64, device=dev) for _ in range(8)]   # would reuse freed blocks
    eng.wait(t)
    assert torch.equal(out, G0)
    R0 = eng.tsqr(st, rhs=rhs).clone()
    outR = torch.zeros_like(R0)
    t = eng.tsqr_submit(strided, outR, rhs=rhs3)
    junk2 = [torch.full((S, st["q"].shape[1]), float("nan"), dtype=torch.float64, device=dev) for _ in range(8)]
    eng.wait(t)
    assert torch.equal(outR, R0)
    del junk, junk2


def test_set_stream_and_blocking_calls_complete_what_is_in_flight(setup):
    import torch

    topo, eng, st, rhs, dev, S = setup
    G0 = eng.gram(st, rhs=rhs).clone()
    R0 = eng.tsqr(st, rhs=rhs).clone()
    outG, outR = torch.zeros_like(G0), torch.zeros_like(R0)
    t1 = eng.gram_submit(st, outG, rhs=rhs)
    t2 = eng.tsqr_submit(st, outR, rhs=rhs)
    s2 = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(s2):
        eng.use_torch_stream()            # fbr_model_set_stream: first waits for both submissions
        torch.cuda.synchronize()
        assert torch.equal(outG, G0) and torch.equal(outR, R0)
        assert torch.equal(eng.gram(st, rhs=rhs), G0)     # and the engine works on the new stream
    eng.use_torch_stream()
    eng.wait()
    # a blocking entry point behind a submission: complete results of both
    outR.zero_()
    t = eng.tsqr_submit(st, outR, rhs=rhs)
    Rm = eng.tsqr_merge(R0, R0)
    assert torch.equal(outR, R0)
    assert float(torch.linalg.norm(Rm.T @ Rm - 2 * (R0.T @ R0)) / torch.linalg.norm(R0.T @ R0)) <= 1e-13
    eng.wait(t)


def test_failed_submission_leaves_nothing_in_flight(setup):
    import torch

    from flobaroid_amd._lib import FbrError

    topo, eng, st, rhs, dev, S = setup
    R0 = eng.tsqr(st, rhs=rhs).clone()
    out = torch.zeros_like(R0)
    t = eng.tsqr_submit(st, out, rhs=rhs)
    bad_rhs = torch.zeros((S * eng.rows, 17), dtype=torch.float64, device=dev)       # more rhs columns than FBR_MAX_RHS
    with pytest.raises(FbrError):
        eng.tsqr_submit(st, torch.zeros((eng.cols + 17, eng.cols + 17), dtype=torch.float64, device=dev), rhs=bad_rhs)
    with pytest.raises(FbrError):
        eng.gram_submit(st, torch.zeros((eng.cols + 17, eng.cols + 17), dtype=torch.float64, device=dev), rhs=bad_rhs)
    eng.wait()
    assert torch.equal(out, R0)
    with pytest.raises(ValueError):
        eng.tsqr_submit({k: v.cpu().numpy() for k, v in st.items()}, out, rhs=rhs.cpu().numpy())   # host data: not a submission
    assert torch.equal(eng.tsqr(st, rhs=rhs), R0)


def test_tsqr_submit_with_weights_and_column_subset(setup):
    import torch

    topo, eng, st, rhs, dev, S = setup
    g = torch.Generator(device=dev).manual_seed(4)
    w = torch.rand((S * eng.rows,), dtype=torch.float64, device=dev, generator=g) + 0.5
    cols = np.arange(3, eng.cols, 2, dtype=np.int32)
    for kw in ({"w": w}, {"cols": cols}, {"w": w, "cols": cols}):
        R0 = eng.tsqr(st, rhs=rhs, **kw).clone()
        outs = [torch.zeros_like(R0), torch.zeros_like(R0)]
        t1 = eng.tsqr_submit(st, outs[0], rhs=rhs, **kw)
        t2 = eng.tsqr_submit(st, outs[1], rhs=rhs, **kw)
        eng.wait(t2)
        assert torch.equal(outs[0], R0) and torch.equal(outs[1], R0)
        # a factor handed on (R_in): streaming over two halves == one call
        half = S // 2
        a = {k: v[:half].contiguous() for k, v in st.items()}
        b = {k: v[half:].contiguous() for k, v in st.items()}
        kwa = dict(kw, **({"w": w[: half * eng.rows].contiguous()} if "w" in kw else {}))
        kwb = dict(kw, **({"w": w[half * eng.rows:].contiguous()} if "w" in kw else {}))
        Ra, Rb = torch.zeros_like(R0), torch.zeros_like(R0)
        ta = eng.tsqr_submit(a, Ra, rhs=rhs[: half * eng.rows].contiguous(), **kwa)
        eng.wait(ta)
        tb = eng.tsqr_submit(b, Rb, rhs=rhs[half * eng.rows:].contiguous(), R_in=Ra, **kwb)
        eng.wait(tb)
        G0 = R0.T @ R0
        assert float(torch.linalg.norm(Rb.T @ Rb - G0) / torch.linalg.norm(G0)) <= 1e-12
