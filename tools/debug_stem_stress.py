"""Two processes share the GPU and hammer the stem forward kernels; every repetition must be bit-identical."""
import ctypes as C, os, sys
import torch, torch.multiprocessing as mp
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

def worker(rank, reps):
    sys.path.insert(0, ROOT)
    dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
    from object_detectors_amd._lib import check, lib
    L = lib(); vp = lambda t: C.c_void_p(t.data_ptr())
    for (n, px) in [(2, 192), (4, 256)]:
        g = torch.Generator().manual_seed(rank)
        img = torch.randn((n, 3, px, px), generator=g).to(dev)
        wp = torch.zeros((32, 32), dtype=torch.bfloat16, device=dev); wp[:, :27] = (torch.randn((32, 27), generator=g) * 0.27).to(dev).bfloat16()
        rows = L.mi355det_stem_rows(n, px, px)
        part = torch.zeros((rows + 64, 2, 32), device=dev); ss = torch.zeros(128, device=dev)
        gam, bet, rm, rv = torch.ones(32, device=dev), torch.zeros(32, device=dev), torch.zeros(32, device=dev), torch.ones(32, device=dev)
        a = torch.empty((n, px, px, 32), dtype=torch.bfloat16, device=dev)
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        ref = None; bad_ss = bad_a = 0; where = []
        for r in range(reps):
            a.fill_(3.0)
            check(L.mi355det_stem_fwd_stats(vp(img), vp(wp), vp(part), n, px, px, st), "s")
            check(L.mi355det_bn_finalize(vp(part), rows, 32, 32, n * px * px, vp(gam), vp(bet), 1e-5, 0.1, vp(rm), vp(rv), vp(ss), st), "f")
            check(L.mi355det_stem_fwd_apply(vp(img), vp(wp), vp(ss), 0.1, vp(a), 32, n, px, px, st), "a")
            torch.cuda.synchronize()
            if ref is None:
                ref = (part[:rows].clone(), ss.clone(), a.clone())
                continue
            if not torch.equal(part[:rows], ref[0]) or not torch.equal(ss, ref[1]):
                bad_ss += 1
                if len(where) < 3:
                    d = (part[:rows] != ref[0]).any(-1).any(-1).nonzero().flatten().tolist()
                    where.append(("partial rows", d[:8], len(d)))
            if not torch.equal(a, ref[2]):
                bad_a += 1
                if len(where) < 6:
                    d = (a != ref[2]).any(-1)                       # [n, h, w]
                    idx = d.nonzero()
                    tiles = sorted({(int(b), int(y) // 8, int(x) // 32) for b, y, x in idx.tolist()})
                    where.append(("a tiles", tiles[:6], len(tiles), "pixels", int(d.sum())))
                    det = []
                    for b, y, x in idx.tolist()[:12]:
                        chs = (a[b, y, x] != ref[2][b, y, x]).nonzero().flatten().tolist()
                        det.append(((b, y, x), "tile-local", (y % 8, x % 32), "channels", chs[:4], len(chs), "got", [round(float(v), 3) for v in a[b, y, x, chs[:3]]],
                                    "ref", [round(float(v), 3) for v in ref[2][b, y, x, chs[:3]]]))
                    print("   detail:", det, flush=True)
        print(f"rank {rank} n {n} px {px}: {reps} reps, stats/ss mismatches {bad_ss}, activation mismatches {bad_a}; {where}", flush=True)

if __name__ == "__main__":
    nproc = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
    ctx = mp.get_context("spawn")
    ps = [ctx.Process(target=worker, args=(r, reps)) for r in range(nproc)]
    [p.start() for p in ps]; [p.join() for p in ps]
