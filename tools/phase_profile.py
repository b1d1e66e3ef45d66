"""Per-phase cycle accounting of the identity kernel.  Needs a timing build of the library:
  hipcc ... -DNPHM_PROF=1 (see nphm_amd/build.py for the flags) -o /tmp/libprof.so ; NPHM_AMD_LIB=/tmp/libprof.so python tools/phase_profile.py
Modes: bf16x3 (all members 3-pass), all-light (NPHM_PROF_LIGHT_TOL=10: every member single-pass), adaptive, f32."""
import os, torch, sys
sys.path.insert(0,"tests"); sys.path.insert(0,".")
import _util as U
from nphm_amd import _lib, reconstruction as R
dev=torch.device("cuda:0")
for prec,code,tol in (("bf16x3",1,None),("all-light",2,"10"),("adaptive",2,None),("adaptive2",3,None)):
    os.environ.pop("NPHM_PROF_LIGHT_TOL",None)
    if tol: os.environ["NPHM_PROF_LIGHT_TOL"]=tol
    net=U.build_identity(device=dev).eval(); net.precision={"all-light":"bf16x3a","adaptive":"bf16x3a","adaptive2":"bf16x3a2"}.get(prec,prec)
    lat=U.sample_latent(0).to(dev)
    res=256
    axes=R.grid_axes(U.MINI,U.MAXI,res)
    ax=[torch.from_numpy(a).to(dev) for a in axes]
    lib=_lib.load()
    packed,state,_=net.prepare_latent(lat[None])
    out=torch.empty(res**3,device=dev)
    binned=os.environ.get("NPHM_PHASE_BINNED","1")!="0"
    ws=R.grid_workspace(dev,res,res,res) if binned else None
    for it in range(2):
        stats=torch.zeros(16,dtype=torch.int64,device=dev)
        rc=lib.nphm_identity_eval_grid(packed.data_ptr(),state.data_ptr(),ax[0].data_ptr(),ax[1].data_ptr(),ax[2].data_ptr(),res,res,res,0,res,25000,1e-7,code,out.data_ptr(),stats.data_ptr(),None if ws is None else ws.data_ptr(),0 if ws is None else ws.numel(),None)
        torch.cuda.synchronize()
    s=stats.cpu().numpy().astype(float)
    nw=s[11]; names=["L0_gemm","sync(active)","gemm","epilogue","member_total","kernel_total","vmcnt_wait(all)","barrier_wait(all)","dma_issue(all)"]
    mw = s[0]/32  # member-waves
    print(prec, "member-waves", mw, "active waves", nw)
    for i,n in enumerate(names): print(f"  {n:14s} {s[2+i]:.4g} ticks  per member-wave {s[2+i]/mw:9.1f}  share of kernel_total {s[2+i]/s[7]:.3f}")
