"""CPU emulation of the dense skip-MLP kernel's operand rounding (tools, not product): how far the three- / two- / one-term
split-f16 products land from an fp64 evaluation of the same network.  Operands are rounded exactly as mlp_kernel.hip does
(weights: hi = rn(w), lo = rn(w - hi); activations in the scaled domain a' = k a: hi by round-toward-zero or to nearest,
lo = rn(x - hi)); accumulation in fp64 (the fp32 accumulation error is common to all modes).
usage: python tools/emulate_mlp_terms.py [trained_def|npm]"""
import sys
import numpy as np

K = 100.0 / np.log(2.0)


def f16(x):
    return x.astype(np.float16).astype(np.float64)


def f16_rtz(x):
    h = x.astype(np.float16)
    over = np.abs(h.astype(np.float64)) > np.abs(x)
    h = np.where(over, np.nextafter(h, np.float16(0)), h)
    return h.astype(np.float64)


def softplus2(d):
    return np.maximum(d, 0) + np.log2(1 + np.exp2(-np.abs(d)))


def run(ws, bs, xyz, cond, terms, rtn_one=True, exact=False):
    """terms[l] in {1,2,3} for hidden GEMM layers; layer 0 / last as the kernel (full precision / three terms)."""
    n_lin = len(ws)
    nl = n_lin - 1
    skip = nl // 2
    d_in = 3 + cond.shape[0]
    x = None
    for l in range(n_lin):
        W, b = ws[l].astype(np.float64), bs[l].astype(np.float64)
        last = l == n_lin - 1
        if l == 0:
            d = (xyz @ W[:, :3].T + cond @ W[:, 3:].T + b) * K
        else:
            ka = x.shape[1]
            scale = 1 / np.sqrt(2) if l == skip else 1.0
            Wa = W[:, :ka] * scale
            add = b.copy()
            if l == skip:
                add = add + (cond @ W[:, ka + 3:].T) * scale
                add = add + xyz @ (W[:, ka:ka + 3] * scale).T
            t = 3 if last else terms.get(l, 3)
            if exact:
                d = x @ Wa.T
            else:
                wh = f16(Wa); wl = f16(Wa - wh)
                if t == 1 and rtn_one:
                    xh = f16(x); xl = 0 * x
                else:
                    xh = f16_rtz(x); xl = f16(x - xh)
                d = xh @ wh.T
                if t >= 2: d = d + xl @ wh.T
                if t >= 3: d = d + xh @ wl.T
            d = d / K + add if last else d + add * K
        x = d if last else softplus2(d)
    return x


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "trained_def"
    rng = np.random.default_rng(0)
    if which == "trained_def":
        s = np.load("tests/golden/trained_def_state.npz")
        ws = [s[f"sd.defDeepSDF.lin{i}.weight"] for i in range(7)]
        bs = [s[f"sd.defDeepSDF.lin{i}.bias"] for i in range(7)]
        g = np.load("tests/golden/trained_def.npz")
        xyz = g["p0_xyz"][0].astype(np.float64)
        anchors = g["p0_anchors"][0].reshape(-1)
        z_ex = s["z_ex"][int(s["pairs"][0][1])]
        # compress mode (deepSDF.py:184-195): cond = [z_ex | compressor(cat(z_id-ish, anchors))]; here any plausible 32-vector
        cond = np.concatenate([z_ex, rng.normal(0, 0.3, 32)]).astype(np.float64)
    else:
        s = np.load("tests/golden/trained_npm_state.npz")
        keys = sorted(k for k in s.files if k.endswith("weight"))
        n = len(keys)
        pre = keys[0][: keys[0].index("lin")]
        ws = [s[f"{pre}lin{i}.weight"] for i in range(n)]
        bs = [s[f"{pre}lin{i}.bias"] for i in range(n)]
        xyz = rng.uniform(-0.5, 0.5, (2048, 3))
        cond = rng.normal(0, 0.1, ws[0].shape[1] - 3)
    nl = len(ws) - 1
    hid = list(range(1, nl))
    ref = run(ws, bs, xyz, cond, {}, exact=True)
    print(which, "out range", np.abs(ref).max())
    def err(terms, **kw):
        return np.abs(run(ws, bs, xyz, cond, terms, **kw) - ref).max()
    print("3-term all        ", err({}))
    print("2-term all        ", err({l: 2 for l in hid}))
    print("1-term all (rtn)  ", err({l: 1 for l in hid}))
    print("1-term all (rtz)  ", err({l: 1 for l in hid}, rtn_one=False))
    for l in hid:
        print(f"  1-term layer {l} only, rest 2:", err({**{m: 2 for m in hid}, l: 1}))
    for l in hid:
        print(f"  1-term all but layer {l} (2):", err({**{m: 1 for m in hid}, l: 2}))


if __name__ == "__main__":
    main()
