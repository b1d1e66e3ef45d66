"""Robustness of the calibrated defaults on another checkpoint than the committed fixture (development tool, GPU):

    python tools/train_synthetic_heads.py --steps 15000 --out gpurun_out/sharper_heads.npz      # ~5 min on an MI355X
    python tools/check_checkpoint.py gpurun_out/sharper_heads.npz
    python tools/check_checkpoint.py EXPERIMENT_DIR/<exp>/checkpoints/checkpoint_epoch_6000.tar     # a reference checkpoint

Per code: the calibrated default against the dense fp32 kernel on the full 128^3 lattice, validate_numerics, and the training
tier against the composite tier with every member (pins the kernels' arithmetic on those weights)."""
import sys, time, numpy as np, torch
sys.path.insert(0, "tests"); sys.path.insert(0, "."); sys.path.insert(0, "tools")
import _util as U
from nphm_amd import reconstruction as R
import nphm_amd
dev = torch.device("cuda:0")
if sys.argv[1].endswith(".npz"):
    ck = np.load(sys.argv[1])                                  # tools/train_synthetic_heads.py
    sd = {k[3:]: torch.from_numpy(ck[k]) for k in ck.files if k.startswith("sd.")}
    codes = torch.from_numpy(ck["codes"]).float().to(dev)
else:
    # the reference's own checkpoint container (training.py:190-201: checkpoint_epoch_{E}.tar = {'decoder_state_dict',
    # 'latent_codes_state_dict': nn.Embedding weights [n_subjects, 1344], optimizer dicts, 'epoch'}), e.g. the released one
    ck = torch.load(sys.argv[1], map_location="cpu")
    sd = ck["decoder_state_dict"]
    codes = ck["latent_codes_state_dict"]["weight"].float().to(dev)
net = U.build_identity(device=dev)
net.load_state_dict(sd, strict=True)                           # the reference's key layout, strict
print("largest |weight|", max(float(v.abs().max()) for k, v in sd.items() if "weight" in k), "codes", tuple(codes.shape))
net.eval()
axes = R.grid_axes(U.MINI, U.MAXI, 128)
worst = 0.0
CODES = [c for c in (0, 9, 23, 41) if c < codes.shape[0]] or [0]
for c in CODES:
    lat = codes[c]
    net.precision, net.prune_tol = "f32", -1.0
    ref = R.evaluate_grid(net, lat, axes, hack_chunk=0)
    net.numerics = "auto"
    t0 = time.perf_counter()
    st = torch.zeros(16, dtype=torch.int64, device=dev)
    got = R.evaluate_grid(net, lat, axes, hack_chunk=0, stats=st)
    torch.cuda.synchronize()
    s = st.cpu().numpy().astype(float) / 128 ** 3
    e = float((got - ref).abs().max()); worst = max(worst, e)
    cal = net.calibration
    print(f"code {c}: max |sdf| {float(ref.abs().max()):.3f}, auto vs dense fp32 {e:.2e}, members {s[0]:.2f} (1-pass {s[15]:.2f}, 2-pass {s[14]:.2f}); picked {cal['precision']} light {cal['light_tol']} mid {cal['mid_tol']} prune {cal['prune_tol']:g} sample err {cal['error']:.2e}")
rep = nphm_amd.validate_numerics(net, codes[CODES], n=1 << 17)
print("validate_numerics:", {k: (f"{v:.3e}" if isinstance(v, float) else v) for k, v in rep.items()})
# fitting tier and training tier against composite
import test_hip_train as T
net.train()
_, xyz, nrm = T._batch(dev, B=len(CODES), N=1000, seed=3)
lat = codes[CODES][:, None, :].contiguous()
ref = T._run(net, "composite", lat, xyz, nrm)
net.numerics = "auto"
for step in range(2):
    out = T._run(net, "hip", lat, xyz, nrm)
    worst_t = sorted(((k, T._rel(out[k], ref[k])) for k in ref), key=lambda kv: -kv[1])[:3]
    print(f"training tier step {step}: next budget {net._train_tol():.1e}", {k: f"{v:.1e}" for k, v in worst_t})
print("worst inference error", worst)
