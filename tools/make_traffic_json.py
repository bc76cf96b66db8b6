"""profiles/<tag>_pmc_step_traffic.txt (tools/pmc_step.sh) -> profiles/traffic.json.

    python tools/make_traffic_json.py profiles/r05x_pmc_step_traffic.txt S spt64 [dtype] > profiles/traffic.json

One entry per roofline op of `hotpath.SPTTrainStep.roofline`, keyed by op AND workload shape
(`<op>@<scene>/<net>`): a bench line of another scene / model finds no entry and prints
`traffic: null` instead of borrowing scene S's bytes.  FETCH_SIZE / WRITE_SIZE are KiB per
launch; FETCH_SIZE is doubled on gfx950 (MI355X_MICROARCH.md, HBM / rocprofv3 section: the
counter ticks once per 128-byte line and is scaled as if lines were 64 bytes).  The attention
entries take the `hi` column of tools/pmc_query.py: the level-1 launches only."""
import json
import re
import sys

path, scene, net = sys.argv[1], sys.argv[2], sys.argv[3]
dtype = sys.argv[4] if len(sys.argv) > 4 else "f32"          # keys of non-f32 captures end in /<dtype>
suffix = "" if dtype == "f32" else f"/{dtype}"
rows = {}
for line in open(path):
    m = re.match(r"(.{48}) (\S+)\s+avg\s+([\d.]+)\s+max\s+([\d.]+)\s+n=(\d+)(?:\s+hi\s+([\d.]+)\s+nhi=(\d+))?", line)
    if not m:
        continue
    name, counter = m.group(1).strip(), m.group(2)
    rows[(name, counter)] = dict(avg=float(m.group(3)), max=float(m.group(4)), n=int(m.group(5)),
                                 hi=float(m.group(6)) if m.group(6) else float(m.group(4)))


def find(sub, counter, col):
    for (name, cn), r in rows.items():
        if sub in name and cn == counter:
            return r[col]
    return None


def entry(label, kernels, col):
    # a kernel may be given as alternatives "a|b": the first one the capture holds
    def pick(k, counter):
        for alt in k.split("|"):
            v = find(alt, counter, col)
            if v is not None:
                return v
        return None
    f = [pick(k, "FETCH_SIZE") for k in kernels]
    w = [pick(k, "WRITE_SIZE") for k in kernels]
    if any(v is None for v in f + w):
        return None
    fetch, write = sum(f), sum(w)
    return {"kernel": label, "fetch_kib": round(fetch, 1), "write_kib": round(write, 1),
            "fetch_correction": 2.0, "bytes": int((2.0 * fetch + write) * 1024),
            "source": f"{path}: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (two passes) over "
                      f"`bench.py --steps 2` at scene {scene} ({net}), KiB per launch; FETCH_SIZE doubled "
                      "per MI355X_MICROARCH.md (exact for wide coalesced reads, uncalibrated for gathers)"}


out = {"_comment": "HBM bytes per launch from rocprofv3 PMC passes over the train step (tools/pmc_step.sh -> "
                   "tools/make_traffic_json.py), read by hotpath.roofline(); keys carry the workload "
                   "(<op>@<scene>/<net>): no entry, no traffic figure"}
spec = {
    "segmax": ("spt::segmax_stream_kernel<true>", ["segmax_stream_kernel"], "avg"),
    "attn_bwd": ("prep + main kernel + per-node reduction of the edge-lane attention backward "
                 "(spt::to::attn_bwd_to_kernel + attn_q_reduce_kernel, or spt::el::attn_bwd_el_kernel + "
                 "attn_kv_reduce_kernel), level-1 launches",
                 ["attn_bwd_to_kernel|attn_bwd_el_kernel", "attn_q_reduce_kernel|attn_kv_reduce_kernel",
                  "attn_bwd_to_prep_kernel|attn_bwd_prep_kernel"], "hi"),
    "attn_fwd": ("spt::mfma::attn_fwd_mfma_kernel, level-1 launches", ["attn_fwd_mfma_kernel"], "hi"),
    "mlp_bwd_pooled": ("spt::fdma::bwd_dma_kernel<64, 128, 8, 2, true, true>", ["bwd_dma_kernel<64, 128"], "avg"),
    "mlp_fwd": ("spt::fmlp::fwd_kernel_x3<16, 8> (or fwd_kernel<16, 8>)", ["fwd_kernel_x3<16, 8>|fwd_kernel<16, 8>"], "avg"),
    # round 5: the top layer with the pool inside (csrc/fused_pool.hip)
    "mlp_fwd_pool": ("spt::fpool::fwd_pool_kernel + pool_apply_kernel", ["fwd_pool_kernel", "pool_apply_kernel"], "avg"),
    "mlp_bwd_pool": ("spt::fpool::pool_bwd_gm_kernel + bwd_pool_kernel", ["pool_bwd_gm_kernel", "bwd_pool_kernel"], "avg"),
    # the north-star kernel on its own (tools/pmc_segmax.sh capture)
    "segmax_standalone": ("spt::segmax_stream_kernel<false>", ["segmax_stream_kernel<false"], "avg"),
}
try:                                   # keep the entries of other workloads / legs
    import os
    old = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                      "profiles", "traffic.json")))
    out.update({k: v for k, v in old.items() if k != "_comment"})
except (OSError, ValueError):
    pass
for op, (label, kernels, col) in spec.items():
    e = entry(label, kernels, col)
    if e is not None:
        out[f"{op}@{scene}/{net}{suffix}"] = e
print(json.dumps(out, indent=1))
