import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from neko_amd import ops
BF = torch.bfloat16; dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
for mode in (3, 1):
    ops.gemm_set_mainloop(mode)
    for (m, n, k) in ((768, 768, 768), (16384, 768, 768)):
        A = torch.randn(m, k, device=dev, generator=g).to(BF)
        Bm = (torch.randn(k, n, device=dev, generator=g) * 0.05).to(BF)
        bias = torch.randn(n, device=dev, generator=g); resid = torch.randn(m, n, device=dev, generator=g)
        ref = A.float() @ Bm.float() + bias + resid
        for variant in ("resid", "noresid"):
            o = torch.full((m, n), float("nan"), device=dev)
            kw = dict(b_kstrided=True, bias=bias, out_f32=o)
            if variant == "resid": kw["resid"] = resid
            ops.gemm(A, Bm, m, n, k, **kw)
            torch.cuda.synchronize()
            r = ref if variant == "resid" else ref - resid
            bad = (o - r).abs() > 0.05
            nb = int(bad.sum())
            print(f"mode {mode} {m}x{n}x{k} {variant}: loop {ops.MAINLOOP_NAMES[ops.gemm_last_mainloop()]} bad {nb}")
            if nb:
                idx = bad.nonzero()[:4000]
                rows, cols = idx[:, 0], idx[:, 1]
                print("   row%256 histogram (32-row bins):", torch.bincount((rows % 256) // 32, minlength=8).tolist())
                print("   col%256 histogram (64-col bins):", torch.bincount((cols % 256) // 64, minlength=4).tolist(), " col%4:", torch.bincount(cols % 4, minlength=4).tolist())
                print("   tile rows:", sorted(set((rows // 256).tolist()))[:20], " tile cols:", sorted(set((cols // 256).tolist())))
                i, j = int(rows[0]), int(cols[0])
                print(f"   first bad ({i},{j}): got {float(o[i, j]):.4f} want {float(r[i, j]):.4f}; got-want {float(o[i,j]-r[i,j]):.4f}; resid there {float(resid[i,j]):.4f}; acc+bias {float(r[i,j]-(resid[i,j] if variant=='resid' else 0)):.4f}")
                # is the wrong value equal to another element's right value?
                d = (r - o[i, j]).abs()
                k2 = int(d.argmin()); print("   nearest correct value at", divmod(k2, n), float(d.min()))
