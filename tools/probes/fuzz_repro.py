import sys, numpy as np, torch
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from test_gpu_fuzz import make_case
from pislam_amd.capi import Context
from pislam_amd.frontend import OrbFrontend
from oracle import orc
seed = int(sys.argv[1])
levels, vstep, rows, pyr, par, opts = make_case(seed)
print(par, opts, levels, vstep, rows, pyr.shape)
dev = torch.device("cuda:0")
def run(o):
    ctx = Context(device=0)
    for k, v in o.items(): ctx.set_option(k, v)
    fe = OrbFrontend(levels, vstep=vstep, rows=rows, ctx=ctx, **par)
    kp, desc, counts = fe.alloc_outputs(len(pyr), dev)
    fe(torch.from_numpy(pyr).to(dev), kp, desc, counts); torch.cuda.synchronize()
    c = counts.cpu().numpy().view(np.uint32); k = kp.cpu().numpy().view(np.uint32)
    bad = 0
    for b in range(len(pyr)):
        okp, odesc, _ = orc.pyramid(pyr[b], levels, fast_threshold=par["fast_threshold"], harris_threshold=par["harris_threshold"], border=par["border"], log_bucket=par["log_bucket_size"], bucket_limit=par["bucket_limit"], words=par["words"])
        m = min(len(okp), par["max_keypoints"])
        if c[b] != len(okp) or not (k[b,:m]==okp[:m]).all():
            bad += 1
            d = np.nonzero(k[b,:m]!=okp[:m])[0]
            print("  slot", b, "count", c[b], len(okp), "first diff", d[:5], [hex(v) for v in k[b,d[:3]]], [hex(v) for v in okp[d[:3]]])
    print(o, "bad" if bad else "ok")
run(opts)
for key in opts:
    o = dict(opts); o[key] = 0 if key != "alias" else 1
    if o != opts: run(o)
