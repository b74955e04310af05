import sys, numpy as np, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from oracle import cnn_gru_oracle as O
from gpu_common import to_t, split_named
from multimodalsignal_amd.runtime import Engine
B, C, K, T, p = 3100, 6, 2, 960, 0.5
params = {k: v.numpy() for k, v in O.init_params(C, K, seed=100 + B).items()}
rs = np.random.RandomState(B * 7 + T)
x = (rs.randn(B, C, T) * (0.5 + rs.rand(1, C, 1)) + rs.randn(1, C, 1)).astype(np.float32)
y = rs.randint(0, K, size=(B,)).astype(np.int64)
dev = torch.device("cuda:0")
eng = Engine(C, K, dev)
pp, bb = split_named(to_t(params))
for k in O.buffer_specs():
    bb.setdefault(k, O.init_buffers()[k])
eng.load_named({**pp, **bb})
xt, yt = torch.as_tensor(x), torch.as_tensor(y)
L1, P1, L2, TP = O.stage_lengths(T)
res = {}
for name, dt in (("f32", torch.float32), ("f64", torch.float64)):
    p2, b2 = split_named(to_t(params, dt))
    for k in bb:
        b2.setdefault(k, bb[k] if "num_batches" in k else bb[k].to(dt))
    loss, grads, st, _ = O.loss_and_grads(p2, b2, xt.to(dt), yt, retain=True, dropout_p=p, seed=1234, step=3)
    res[name] = (grads, st)
    print(name, "oracle done", flush=True)
b = eng.forward(xt.to(dev), yt.to(dev), training=True, dropout_p=p, seed=1234, step=3)
eng.backward(b); torch.cuda.synchronize()
dy2 = eng.region("DY2", torch.float32, (B, L2, 32)).cpu().numpy().transpose(0, 2, 1)
for name in ("f32", "f64"):
    ref = res[name][0]["stage/bn2"].numpy().astype(np.float64)
    d = np.abs(dy2 - ref); sc = np.abs(ref).max()
    bad = np.argwhere(d > 1e-4 * sc)
    print(name, "max|ref|", sc, "n bad", len(bad), "of", d.size, "max err", d.max() / sc)
    for idx in bad[:12]:
        bi, ci, ti = idx
        print("   at", idx, "hip", dy2[bi, ci, max(ti-2,0):ti+3], "ref", ref[bi, ci, max(ti-2,0):ti+3])
r32, r64 = res["f32"][0]["stage/bn2"].numpy().astype(np.float64), res["f64"][0]["stage/bn2"].numpy()
d = np.abs(r32 - r64); print("fp32 oracle vs fp64 oracle: n bad", int((d > 1e-4 * np.abs(r64).max()).sum()), "max", d.max() / np.abs(r64).max())
for k in ("cnn_encoder.4.weight", "cnn_encoder.0.weight"):
    g = eng.named_param_views(eng.grads)[k].cpu().numpy().astype(np.float64)
    a, c = res["f32"][0][k].numpy().astype(np.float64), res["f64"][0][k].numpy()
    print(k, "hip vs f64", np.abs(g - c).max() / np.abs(c).max(), " f32 oracle vs f64", np.abs(a - c).max() / np.abs(c).max(), " hip vs f32 oracle", np.abs(g - a).max() / np.abs(c).max())
