"""Shared helpers for the -m gpu parity tests: run the HIP path through the C ABI and
collect, stage by stage, its deviation from the CPU oracle."""
import numpy as np
import torch

from oracle import cnn_gru_oracle as O


def to_t(d, dtype=torch.float32):
    out = {}
    for k, v in d.items():
        t = torch.as_tensor(v)
        out[k] = t if "num_batches" in k else t.to(dtype)
    return out


def split_named(named):
    params = {k: v for k, v in named.items() if "running" not in k and "num_batches" not in k}
    buffers = {k: v for k, v in named.items() if k not in params}
    return params, buffers


def rel_err(a, ref):
    a = np.asarray(a, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    if ref.size == 0:
        return 0.0
    scale = max(np.abs(ref).max(), 1e-12)
    return float(np.abs(a - ref).max() / scale)


def routed_dz(dp, code, L):
    """dz = dL/d(bn output) (B, L, CH) from dP (B, P, CH) and the forward pass's pooling decisions (B, P, CH/4) bytes, 2 bits
    per channel: 0/1/2 = left/centre/right candidate won and is positive, 3 = no gradient (include/msig.h WS_POOLC*)."""
    B, P, CH = dp.shape
    win = (code[:, :, :, None] >> (2 * np.arange(4, dtype=np.uint8))[None, None, None, :]) & 3     # (B, P, CH/4, 4)
    win = win.reshape(B, P, CH)
    dz = np.zeros((B, L, CH), dtype=np.float64)
    ph = np.arange(P)
    for k in range(3):
        t = 2 * ph - 1 + k
        ok = (t >= 0) & (t < L)
        contrib = np.where(win == k, dp, 0.0)[:, ok, :]
        dz[:, t[ok], :] += contrib          # for a fixed candidate index the target positions are distinct
    return dz


def run_case(engine, named, x, y, dropout_p=0.0, seed=0, step=0, check_backward=True):
    """Returns {stage: (relative max error, tolerance)} for forward (+ backward) stages."""
    import multimodalsignal_amd._lib as L
    C, K = engine.C, engine.K
    B, _, T = x.shape
    L1, P1, L2, TP = O.stage_lengths(T)
    params, buffers = split_named(to_t(named))
    for k, spec in O.buffer_specs().items():
        if k not in buffers:
            buffers[k] = O.init_buffers()[k]
    engine.load_named({**params, **buffers})
    xt, yt = torch.as_tensor(x), torch.as_tensor(y)
    dev = engine.device
    b = engine.forward(xt.to(dev), yt.to(dev), training=True, dropout_p=dropout_p, seed=seed, step=step)
    torch.cuda.synchronize()
    # fp64 oracle: tells how far fp32 itself is from exact arithmetic
    fw = dict(dropout_p=dropout_p, seed=seed, step=step)
    p64, b64 = split_named(to_t(named, torch.float64))
    for k in buffers:
        b64.setdefault(k, buffers[k] if "num_batches" in k else buffers[k].double())
    loss64, grads64, st64, _ = O.loss_and_grads(p64, b64, xt.double(), yt, retain=True, **fw)
    # MaxPool's argmax is a discrete decision: where two candidates of a window agree to within fp32 resolution, which of
    # them is "the" maximum differs between any two fp32 implementations (one such window in ~2e7 at B=3100 x T=960), and
    # the gradient then lands one or two positions away.  Only for such near-ties the oracle adopts the decision of the
    # HIP path (recomputed here from ITS conv output and BN constants, as bn_relu_pool / pool_bn_bwd_pass1 do:
    # z = fma(y, scale, shift), first maximum wins); everywhere else the oracle's own argmax stands, so a wrong tie
    # rule or a wrong window still fails.  The number of adopted decisions is reported and bounded.
    choice, n_adopted = {}, 0
    for stage, yname, sname, CH, Lc in (("pool1", "Y1", "BN1_STAT", 16, L1), ("pool2", "Y2", "BN2_STAT", 32, L2)):
        yh = engine.region(yname, torch.float32, (B, Lc, CH)).cpu().double().permute(0, 2, 1)
        stt = engine.region(sname, torch.float32, (4, CH)).cpu().double()
        zh = (yh * stt[2][None, :, None] + stt[3][None, :, None]).float()            # fma in fp32: exact product, one rounding
        ch_hip = O.first_argmax(O.pool_windows(torch.clamp_min(zh, 0)))
        win = O.pool_windows(torch.clamp_min(st64["bn" + stage[-1]].detach(), 0))
        ch_ref = O.first_argmax(win)
        top = win.max(dim=3).values
        hip_val = win.gather(3, ch_hip.to(torch.int64)[..., None]).squeeze(3)
        near = (ch_hip != ch_ref) & ((top - hip_val) <= 4e-6 * torch.clamp_min(top.abs(), 1e-3))   # HIP's pick is within fp32 noise of the maximum
        n_adopted += int(near.sum())
        choice[stage] = torch.where(near, ch_hip, ch_ref)
    if n_adopted:
        loss64, grads64, st64, _ = O.loss_and_grads(p64, b64, xt.double(), yt, retain=True, pool_choice=choice, **fw)
    # the fp32 oracle routes through the same windows, so that `own` below measures arithmetic, not its own near-tie flips
    loss, grads, st, nb = O.loss_and_grads(params, buffers, xt, yt, retain=True, pool_choice=choice, **fw)
    R = lambda name, shape, dtype=torch.float32: engine.region(name, dtype, shape).cpu().numpy()
    rep = {"pool_near_ties_adopted": (float(n_adopted), 8.0)}

    def put(name, got, key, tol):
        ref64 = st64[key].detach().numpy() if isinstance(key, str) else key
        rep[name] = (rel_err(got, ref64), tol)

    put("gate_s", R("GATE_S", (B, C)), "gate_s", 2e-6)
    put("conv1", R("Y1", (B, L1, 16)).transpose(0, 2, 1), "conv1", 5e-6)
    put("pool1", R("P1", (B, P1, 16)).transpose(0, 2, 1), "pool1", 1e-5)
    put("conv2", R("Y2", (B, L2, 32)).transpose(0, 2, 1), "conv2", 1e-5)
    put("pool2", R("P2", (B, TP, 32)).transpose(0, 2, 1), "pool2", 2e-5)
    put("gru_l0", R("H0", (B, TP, 128)), "gru_l0", 5e-5)
    put("gru_l1_fwd", R("H1", (B, TP, 64)), "gru_l1_fwd", 1e-4)
    put("feat", R("FEAT", (B, 128)), "feat", 1e-4)
    put("cls_hidden", R("HID", (B, 64)), "cls_hidden", 1e-4)
    put("logits", R("LOGITS", (B, K)), "logits", 1e-4)
    rep["loss"] = (abs(float(R("LOSS", (4,))[0]) - float(loss64)) / max(abs(float(loss64)), 1e-6), 1e-5)
    bn = engine.bn_state.cpu().numpy()
    for i, (k, sl) in enumerate((("cnn_encoder.1.running_mean", slice(0, 16)), ("cnn_encoder.1.running_var", slice(16, 32)),
                                 ("cnn_encoder.5.running_mean", slice(32, 64)), ("cnn_encoder.5.running_var", slice(64, 96)))):
        rep[k] = (rel_err(bn[sl], nb[k].numpy()), 1e-5)
    if not check_backward:
        return rep, None
    engine.backward(b)
    torch.cuda.synchronize()
    g64 = lambda k: grads64[k].numpy()
    put("d_logits", R("DLOGITS", (B, K)), g64("stage/logits"), 1e-4)
    put("d_feat", R("DFEAT", (B, 128)), g64("stage/feat"), 2e-4)
    put("d_gru_l0", R("DH0", (B, TP, 128)), g64("stage/gru_l0_dropped"), 5e-4)
    dx0 = R("DX0", (2, B, TP, 32))
    put("d_pool2", (dx0[0] + dx0[1]).transpose(0, 2, 1), g64("stage/pool2"), 1e-3)
    # WS_DY2 holds dL/d(bn2 output) (dP2 routed through the forward pass's pooling decisions WS_POOLC2; the BatchNorm-backward
    # second pass of stage 2 is fused into the conv2 backward kernels).  Stage 1's dz is never stored: conv1_bwd routes WS_DP1
    # through WS_POOLC1 on the fly — rebuilt here the same way, and the stage-2 tensor is cross-checked against its own codes.
    dz2 = R("DY2", (B, L2, 32))
    put("d_bn2", dz2.transpose(0, 2, 1), g64("stage/bn2"), 1e-3)
    np.testing.assert_array_equal(dz2, routed_dz((dx0[0] + dx0[1]).astype(np.float64), R("POOLC2", (B, TP, 8), torch.uint8), L2).astype(np.float32))
    dp1 = R("DP1", (B, P1, 16))
    put("d_pool1", dp1.transpose(0, 2, 1), g64("stage/pool1"), 1e-3)
    put("d_bn1", routed_dz(dp1.astype(np.float64), R("POOLC1", (B, P1, 4), torch.uint8), L1).transpose(0, 2, 1), g64("stage/bn1"), 1e-3)
    if "stage/gate_s" in grads64:
        put("d_gate_s", R("DS", (B, C)), g64("stage/gate_s"), 1e-3)
    gviews = engine.named_param_views(engine.grads)
    for k in L.PARAM_KEYS:
        # tolerance: 20x the fp32-vs-fp64 disagreement of the oracle itself, floored at 1e-3 relative
        own = rel_err(grads[k].numpy(), g64(k))
        rep["grad/" + k] = (rel_err(gviews[k].cpu().numpy(), g64(k)), max(1e-3, 20 * own))
    return rep, (loss64, grads64)


def format_report(rep):
    lines = []
    for k, (e, tol) in rep.items():
        lines.append(f"{'FAIL' if not (e <= tol) else 'ok  '} {k:45s} err={e:.3e} tol={tol:.1e}")
    return "\n".join(lines)


def failures(rep):
    return [k for k, (e, tol) in rep.items() if not (e <= tol)]
