"""Benchmark of the SPT hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--scene S|T|D|R]

One "step" = one pass of the hot path over one synthetic
NAG batch already resident in HBM.  N > 1: one process per GPU (torchrun env),
every rank owns a different scene of the same shape (scenes shard data
parallel, "weak" scaling), the only collective is the gradient all-reduce of
the model parameters (RCCL).  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=None,
                   help="ranks = GPUs of this node (default: WORLD_SIZE when a launcher set it, else 1)")
    p.add_argument("--steps", type=int, default=10)
    p.add_argument("--warmup", type=int, default=3)
    p.add_argument("--settle", type=float, default=3.0,
                   help="seconds of untimed steps before the warm-up steps (GPU clock ramp)")
    p.add_argument("--scene", default="S")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-preprocess", action="store_true")
    p.add_argument("--no-f32-exact", action="store_true",
                   help="skip the extra steps that time the f32-matrix-pipe and the bf16 variants of the step")
    p.add_argument("--cpu-scale", type=float, default=None)
    p.add_argument("--cpu-all-cores", action="store_true",
                   help="also probe the CPU baseline with one thread per core (minutes on a 256-core box)")
    p.add_argument("--stages", default="all")
    p.add_argument("--mode", default="train", choices=["train", "infer", "panoptic", "iteration"],
                   help="train: fwd+loss+bwd+AdamW (default, BASELINE cfg #2); infer: forward only "
                        "(cfg #3, use --scene D); panoptic: + edge-affinity head and loss (cfg #5); iteration: "
                        "the on-device transform chain on a raw NAG + the train step, timed together (use --scene T)")
    p.add_argument("--dtype", default="f32", choices=["f32", "bf16", "f32-exact"],
                   help="matrix-pipe precision (superpoint_transformer_amd.precision); f32 = the "
                        "reference's shipped `precision: 32`, bf16 = its bf16 option (cfg #2)")
    p.add_argument("--model", default="spt64", choices=["spt64", "spt128"],
                   help="spt128 = the KITTI-360 width (cfg #4)")
    p.add_argument("--graph", default="random", choices=["random", "local"],
                   help="superpoint graph of the synthetic NAG: random = uniformly drawn endpoints (no "
                        "locality: the stress case, default); local = kNN on the segment centroids (SURVEY 8d)")
    p.add_argument("--order", default="storage", choices=["storage", "morton", "grouped"],
                   help="node order: storage = shuffled (default); morton = levels 1-2 along a Morton curve; "
                        "grouped = the layout of transforms.MortonOrder (levels sorted by parent then curve, "
                        "level-0 points grouped by superpoint)")
    p.add_argument("--scene-mix", action="store_true",
                   help="N > 1: every rank draws its scene size from the S3DIS area distribution (SCENE_MIX) "
                        "instead of all ranks owning scenes of one shape - shows the imbalance loss of "
                        "data-parallel scene sharding (SURVEY 8e); value = all ranks' points / slowest rank")
    p.add_argument("--no-local", action="store_true",
                   help="skip the extra steps on the spatially local superpoint graph (ms_per_step_local)")
    p.add_argument("--no-train-batch", action="store_true",
                   help="skip the train-batch companion (scene T: ms_per_step_T eager + captured, "
                        "ms_per_iteration_T) of the default line")
    p.add_argument("--train-batch-child", action="store_true", help=argparse.SUPPRESS)
    p.add_argument("--capture", action="store_true",
                   help="replay the step as a captured graph (hipGraph through torch.cuda.CUDAGraph): forward + "
                        "loss + backward (+ AdamW at N = 1) of the fixed batch, captured once after the warm-up")
    p.add_argument("--rebuild-csr", action="store_true",
                   help="worst case: ignore the NAG's stored level CSR (nag[i+1].sub) and rebuild every "
                        "CSR view with the device sort each step")
    return p.parse_args()


# Relative sizes of the six S3DIS areas (points after the reference's voxelisation: Area 1-6 =
# 0.96, 1.04, 0.41, 0.95, 1.73, 0.90 of the mean; Area 5, the validation fold of cfg #2, is the
# big one).  `--scene-mix` gives rank r the scene scaled by SCENE_MIX[r % 6].
SCENE_MIX = (0.96, 1.04, 0.41, 0.95, 1.73, 0.90)


def cpu_baseline(scene, scale, all_cores=False):
    """The same step (SPT-64 fwd + loss + bwd) on this box's host cores through
    the CPU oracle (oracle/spt_model.py: the reference's call graph on
    torch-CPU f32 tensors, scatter ops restated from torch_scatter) on a
    bounded sample of the workload.  A reported baseline, not the target."""
    import copy
    from oracle import spt_model as OM
    from superpoint_transformer_amd import hotpath
    from superpoint_transformer_amd.synthetic import SCENES, make_nag
    n0_full = SCENES[scene][0]
    # two thread counts, the better one is the baseline: 128+ threads on sub-millisecond torch-CPU
    # ops have been slower than 32 (round-1 figure: 0.012 Mpts/s), ALL cores are timed next to it
    cores = os.cpu_count() or 1
    if scale is None:
        scale = min(1.0, max(0.05 * n0_full, 150_000) / n0_full)   # >= 5 % of the workload
    nag = make_nag(scene, seed=1234, device="cpu", scale=scale)
    n = nag.num_points
    torch.manual_seed(0)
    model = hotpath.SPTSegmenter(**hotpath.spt64_config(nag[0]["x"].shape[1],
                                                        nag[1]["edge_attr"].shape[1]))
    g = torch.Generator().manual_seed(5)
    labels = [torch.randint(0, hotpath.NUM_CLASSES, (n[i],), generator=g) for i in (1, 2)]
    loss_fn = torch.nn.CrossEntropyLoss()

    def step():
        outs = OM.spt_forward(model.net, nag.levels, dtype=torch.float32, keep_graph=True)
        logits = [h(x) for h, x in zip(model.head, outs)]
        loss = sum(l * loss_fn(lg, y) for l, lg, y in zip((1.0, 50.0), logits, labels))
        model.zero_grad(set_to_none=True)
        loss.backward()

    def timed(budget_s):
        step()                                    # warm-up (allocator, thread pool)
        times = []
        while len(times) < 5:
            t0 = time.perf_counter()
            step()
            times.append(time.perf_counter() - t0)
            if sum(times) > budget_s and len(times) >= 2:
                break
        return sorted(times)[len(times) // 2], len(times)

    by_threads, notes = {}, {}
    torch.set_num_threads(min(32, cores))
    by_threads[min(32, cores)] = timed(30.0)
    if cores > 32 and all_cores:
        # all cores (opt-in: `--cpu-all-cores`).  Measured on the 256-core box of this round
        # (profiles/r05i_bench_sceneS.json): ONE step of this sample with 256 threads took 330.7 s
        # = 0.0023 Mpoints/s against 0.049 with 32 - hundreds of threads on sub-millisecond
        # torch-CPU ops spend their time in the thread pool's hand-shakes.  A probing step slower
        # than twice the 32-thread step is recorded as such and not repeated.
        torch.set_num_threads(cores)
        t0 = time.perf_counter()
        step()
        probe = time.perf_counter() - t0
        if probe <= 2.0 * by_threads[min(32, cores)][0]:
            by_threads[cores] = timed(15.0)
        else:
            notes[str(cores)] = f"one step took {probe:.1f} s ({n[0] / probe / 1e6:.4f} Mpoints/s): not repeated"
        torch.set_num_threads(min(32, cores))
    elif cores > 32:
        notes[str(cores)] = ("not timed in the default run: one step of this sample with all 256 cores of the "
                             "round-5 box took 330.7 s = 0.0023 Mpoints/s (profiles/r05i_bench_sceneS.json; "
                             "`--cpu-all-cores` repeats the probe)")
    best = min(by_threads, key=lambda k: by_threads[k][0])
    dt, nrep = by_threads[best]
    return {"value": round(n[0] / dt / 1e6, 4), "unit": "Mpoints/s",
            "cores": cores, "threads_used": best, "kind": "port",
            "by_threads": {**{str(k): round(n[0] / v[0] / 1e6, 4) for k, v in by_threads.items()}, **notes},
            "sample": f"scene {scene} scaled x{scale:.4g}: N=({n[0]},{n[1]},{n[2]}), median of "
                      f"{nrep} reps of SPT-64 fwd+loss+bwd on torch-CPU f32 via oracle/spt_model.py, "
                      f"timed with {sorted(by_threads)} threads (`value` = the fastest; all {cores} cores: see by_threads)",
            "cut_pursuit": "not timed - dependency unavailable (the reference's CPU partition is "
                           "an un-vendored C++ submodule; out of scope per SURVEY 8)"}


PRE_CFG = {  # configs/datamodule/semantic/{s3dis,dales}.yaml: voxel, knn k, knn r
    "S": (0.03, 45, 2.0), "T": (0.03, 45, 2.0), "R": (0.03, 45, 2.0), "D": (0.10, 25, 10.0)}
# synthetic cloud geometry: indoor rooms (3 m patches in a 50 x 50 x 5 m block) vs an aerial
# tile (20 m patches of ground / roofs / facades over 350 x 350 x 30 m: ~12 M voxels of 10 cm)
PRE_GEOM = {"D": dict(patch=20.0, extent=(350.0, 350.0, 30.0))}


def preprocess_leg(scene, n_points, dev, reps=5):
    """Preprocessing half of the metric: KNN (utils/neighbors.py:51-123) +
    PointFeatures' geometric features (utils/geometry.py:80-126) on a synthetic
    voxelised cloud of the scene's size, inputs resident in HBM."""
    from superpoint_transformer_amd import neighbors as NB
    from superpoint_transformer_amd.synthetic import make_voxel_cloud
    voxel, k, r = PRE_CFG.get(scene, PRE_CFG["S"])
    pos = make_voxel_cloud(n_points, voxel=voxel, seed=4321, device=dev, **PRE_GEOM.get(scene, {}))
    n_points = pos.shape[0]          # one point per voxel: slightly fewer than requested

    def step():
        nb, _ = NB.knn_1(pos, k, r)
        return NB.geometric_features(pos, nb, k_min=1)

    step()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    t_knn, t_geof, t_all = [], [], []
    for _ in range(reps):
        t0 = time.perf_counter()
        ev[0].record()
        nb, _ = NB.knn_1(pos, k, r)
        ev[1].record()
        NB.geometric_features(pos, nb, k_min=1)
        ev[2].record()
        torch.cuda.synchronize()
        t_all.append(time.perf_counter() - t0)
        t_knn.append(ev[0].elapsed_time(ev[1]))
        t_geof.append(ev[1].elapsed_time(ev[2]))
    # median over the repetitions: one of them occasionally takes 5x (allocator / clocks)
    med = lambda v: sorted(v)[len(v) // 2]
    dt_two = med(t_all)
    # the same three outputs (neighbours, distances, features) from ONE call: the kNN kernel sums
    # the neighbourhoods' moments itself (neighbors.knn_1_features -> spt_grid_knn_geof_f32)
    del nb
    NB.knn_1_features(pos, k, r, k_min=1)
    torch.cuda.synchronize()
    t_fused = []
    for _ in range(reps):
        t0 = time.perf_counter()
        NB.knn_1_features(pos, k, r, k_min=1)
        torch.cuda.synchronize()
        t_fused.append(time.perf_counter() - t0)
    dt = med(t_fused)
    # SURVEY 8(d) algorithmic bytes: kNN >= 12 + K (8 + 4) B per point (the candidate scan itself is
    # served by L2 / LDS: the kernel is VALU / LDS-issue bound, its HBM fraction is small BY DESIGN -
    # the VALU-busy share of a PMC capture is the figure that says how close it is to its bound);
    # eigenfeatures (k + 1) (4 + 12) + 44 B per point (gathered positions)
    knn_b, geof_b = n_points * (12 + 12 * k), n_points * ((k + 1) * 16 + 44)
    knn_gbs = knn_b / (med(t_knn) * 1e-3) / 1e9
    geof_gbs = geof_b / (med(t_geof) * 1e-3) / 1e9
    pmc = _preprocess_pmc(scene)
    roof = {"knn": {"kernel": "spt::knn_cell_kernel (+ grid build, leftovers)", "bound": "valu+lds",
                    "bound_note": "per-wave VALU share of a PMC capture x 2 waves per SIMD; the two "
                                  "candidate passes saturate the CU's LDS pipe (DESIGN 7.3 item 8)",
                    "bytes_per_launch": int(knn_b), "achieved": round(knn_gbs, 1), "unit": "GB/s",
                    "frac_hbm": round(knn_gbs / HBM_PEAK_GBS, 4),
                    "valu_busy": (pmc.get("knn_valu_busy_r05") or pmc.get("knn_valu_busy")) if pmc else None,
                    "valu_busy_source": (pmc.get("source_r05") if pmc.get("knn_valu_busy_r05")
                                         else pmc.get("source")) if pmc else None},
            "geof": {"kernel": "spt::point_geof_dense_kernel", "bound": "hbm",
                     "bytes_per_launch": int(geof_b), "achieved": round(geof_gbs, 1), "unit": "GB/s",
                     "peak": HBM_PEAK_GBS, "frac": round(geof_gbs / HBM_PEAK_GBS, 4)}}
    fused_b = n_points * (12 + 12 * (k + 1) + 44)
    roof["knn_geof"] = {"kernel": "spt::knn_cell_kernel<true> (+ grid description, grid build, leftovers): "
                                  "the one-call entry `value` is quoted on",
                        "bound": "valu+lds", "bytes_per_launch": int(fused_b),
                        "valu_busy": pmc.get("knn_geof_valu_busy") if pmc else None,
                        "valu_busy_source": pmc.get("source_r05") if pmc else None,
                        "achieved": round(fused_b / dt / 1e9, 1), "unit": "GB/s",
                        "frac_hbm": round(fused_b / dt / 1e9 / HBM_PEAK_GBS, 4),
                        "bound_note": "as `knn`; the moment sums and the 3 x 3 eigenproblems add f64 "
                                      "vector work (+2.6 ms at scene S), no HBM traffic beyond the 44 B "
                                      "feature row per point"}
    return {"value": round(n_points / dt / 1e6, 3), "unit": "Mpoints/s",
            "workload": f"knn_1(k={k}, r={r}) + geometric_features on {n_points} synthetic "
                        f"voxelised-surface points ({voxel} m lattice): neighbours, distances and "
                        "features out of one call (neighbors.knn_1_features)",
            "ms_total": round(dt * 1e3, 3),
            "two_calls": {"value": round(n_points / dt_two / 1e6, 3), "ms_knn": round(med(t_knn), 3),
                          "ms_geof": round(med(t_geof), 3), "ms_total": round(dt_two * 1e3, 3),
                          "note": "knn_1, then geometric_features on the stored index rows (the "
                                  "reference's two transforms as two entries; the roofline block "
                                  "below describes these two kernels)"},
            "reps": reps, "roofline": roof}


def _preprocess_pmc(scene):
    """VALU-busy share of the kNN cell kernel from a committed PMC capture of this scene's
    settings (profiles/traffic.json key ``preprocess@<scene>``), or None."""
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            return json.load(f).get(f"preprocess@{scene}")
    except (OSError, ValueError):
        return None


def cpu_preprocess_baseline(scene, n_sample):
    """Same leg on the host cores through the CPU twin (oracle/cpu/spt_cpu.cpp: the same exact
    grid kNN contract + the eigenfeatures in f64, OpenMP over all cores) on a bounded sample of
    the same cloud - the same ALGORITHM class on the CPU, not an O(N^2) strawman."""
    from oracle import cpu_twin as T
    from superpoint_transformer_amd.synthetic import make_voxel_cloud
    voxel, k, r = PRE_CFG.get(scene, PRE_CFG["S"])
    threads = os.cpu_count() or 1
    # bounded to ~25 s of host time: the twin runs ~2 300 points / s / core at these settings
    # (14.8 M points in 25.5 s on 256 cores); the full scene where the cores allow it
    n_sample = min(int(n_sample), max(100_000, 58_000 * threads))
    pos = make_voxel_cloud(n_sample, voxel=voxel, seed=4321, device="cpu", **PRE_GEOM.get(scene, {}))
    n_sample = pos.shape[0]
    t0 = time.perf_counter()
    nb, _ = T.knn_1(pos, k, r, threads=threads)
    t1 = time.perf_counter()
    T.point_geof(pos, nb, k_min=1, threads=threads)
    t2 = time.perf_counter()
    return {"value": round(n_sample / (t2 - t0) / 1e6, 4), "unit": "Mpoints/s",
            "cores": os.cpu_count(), "threads_used": threads, "kind": "port",
            "s_knn": round(t1 - t0, 3), "s_geof": round(t2 - t1, 3),
            "sample": f"{n_sample} points of the same synthetic cloud generator: exact grid kNN "
                      f"(k={k}, r={r}) + eigenfeatures via oracle/cpu/libspt_cpu.so (OpenMP)"}


def train_batch_leg(dev):
    """The train-batch companion in a CHILD process (same GPU, the parent idle): whatever happens
    in it - a capture the runtime refuses, a crash - costs the default line its `train_batch`
    object, never the headline."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--train-batch-child"]
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    except subprocess.TimeoutExpired:
        return {"error": "the train-batch child timed out after 900 s"}
    for ln in reversed(r.stdout.splitlines()):
        if ln.startswith('{"scene"'):
            return json.loads(ln)
    tail = (r.stderr or "").strip().splitlines()[-3:]
    return {"error": f"the train-batch child exited with {r.returncode}: {' | '.join(tail)[:400]}"}


def _train_batch_leg(dev, steps=100):
    """The regime the reference TRAINS in, next to the 15 M-point headline: scene T = one S3DIS
    train batch (4 clouds, 1.2 M points after sampling; batch_size / sample caps of
    configs/datamodule/semantic/s3dis.yaml:101-104, default.yaml:79-80).  Three figures:
    the eager step (`ms_per_step_T_eager`), the same step replayed as ONE captured graph
    (`ms_per_step_T`: hotpath.SPTTrainStep.capture - forward + CE + backward + AdamW, per-batch
    CSR builds inside), and a whole training ITERATION as the reference runs it
    (`ms_per_iteration_T`: src/datamodules/base.py:341-380 - the on-device transform chain on a raw
    batch, then the step; dynamic shapes, so eager)."""
    from superpoint_transformer_amd import csr as _csr, hotpath, ops
    from superpoint_transformer_amd.synthetic import SCENES, make_nag, make_raw_nag
    paused = ops.pause_timers(True)               # the headline's timers stay the headline's
    out = {"scene": "T", "sizes": list(SCENES["T"])}
    try:
        nag = make_nag("T", seed=1234, device=dev)
        path = hotpath.build(nag, dev, mode="train", model="spt64")

        def timed(p, n):
            for _ in range(10):
                p.step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                p.step()
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / n * 1e3

        out["ms_per_step_T_eager"] = round(timed(path, steps), 4)
        # (the iteration leg runs BEFORE the capture: its varying shapes then meet an allocator that
        # holds no graph-private pool)
        torch.cuda.empty_cache()
        raw = make_raw_nag("T", seed=1234, device=dev)
        it = hotpath.build(raw, dev, mode="iteration", model="spt64")
        for _ in range(40):                       # (sampled sizes vary: let the allocator see them)
            it.step()
        it.reset_kernel_timers()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n_it = max(steps // 2, 10)
        for _ in range(n_it):
            it.step()
        torch.cuda.synchronize()
        out["ms_per_iteration_T"] = round((time.perf_counter() - t0) / n_it * 1e3, 4)
        r = it.roofline(HBM_PEAK_GBS)
        out["ms_transform_chain"] = r.get("ms_transform_chain")
        out["iteration"] = {"raw_points": int(raw.num_points[0]),
                            "sizes_after_chain": r.get("sizes_after_chain"),
                            "note": "raw batch of 1.2 M points as stored -> the chain keeps "
                                    f"{r.get('sizes_after_chain', [None])[0]} level-0 points, which is what the "
                                    "model steps on; chain + step timed together, eager"}
        del it, raw
        torch.cuda.empty_cache()
        try:
            path.capture()
            out["ms_per_step_T"] = round(timed(path, steps), 4)
            out["captured"] = "forward + CE loss + backward + AdamW of the fixed-cap batch in one graph"
            _csr.verify_adopted(block=True)       # the captured view checks' verdicts, outside the graph
        except Exception as e:                    # a capture that fails is reported, not hidden
            out["ms_per_step_T"] = out["ms_per_step_T_eager"]
            out["captured"] = f"capture failed ({type(e).__name__}: {str(e)[:200]}): eager figure"
        out["Mpoints_per_s_T"] = round(nag.num_points[0] / out["ms_per_step_T"] / 1e3, 2)
        out["workload"] = path.describe("T", SCENES["T"])
        del path, nag
        torch.cuda.empty_cache()
    finally:
        ops.pause_timers(paused)
    return out


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(args):
    """``python bench.py --gpus N`` without a launcher: start N ranks ourselves, one
    process per GPU (the reference's ``ddp_spawn``, configs/trainer/ddp.yaml:8-13),
    through torch.distributed.run on the loopback address.  Rank 0's JSON line is
    the child's stdout, passed through unchanged."""
    import subprocess
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
           f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    raise SystemExit(subprocess.call(cmd, env=env))


_T0 = time.perf_counter()


def _log(msg):
    """Progress on stderr (stdout carries the one JSON line): where a slow run spends its time."""
    print(f"[bench {time.perf_counter() - _T0:7.1f}s] {msg}", file=sys.stderr, flush=True)


def main():
    args = parse()
    if args.train_batch_child:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs an MI355X (no CPU fallback path exists)")
        torch.cuda.set_device(0)
        from superpoint_transformer_amd import precision
        precision.set_matrix_precision("f32")
        print(json.dumps(_train_batch_leg(torch.device("cuda", 0))))
        return
    if args.gpus is None:
        # under torchrun / an external launcher WORLD_SIZE decides; alone, one GPU
        args.gpus = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        launch_ranks(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        # only an EXPLICIT --gpus that contradicts the launcher is an error
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback path exists)")
    # test hook: SPT_BENCH_SHARE_GPU=1 runs every rank on cuda:0 over gloo, to exercise
    # the N > 1 code path on a 1-GPU box (RCCL refuses two ranks on one device)
    share = os.environ.get("SPT_BENCH_SHARE_GPU") == "1"
    if share:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    # SPT_FORCE_COLLECTIVES=1: initialise the process group (RCCL) and run the gradient all-reduce
    # even with ONE rank - proves nothing about scaling, but executes the collective path on a
    # one-GPU box (tests/test_rccl_gpu.py)
    force_dist = os.environ.get("SPT_FORCE_COLLECTIVES") == "1"
    if world > 1 or force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if force_dist and world == 1:
            os.environ.setdefault("MASTER_PORT", str(_free_port()))
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        if not share and torch.cuda.device_count() < world:
            raise SystemExit(f"bench.py: --gpus {world} needs {world} visible devices, "
                             f"found {torch.cuda.device_count()}")
        if share:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
        assert dist.get_world_size() == args.gpus

    from superpoint_transformer_amd import hotpath
    from superpoint_transformer_amd.synthetic import SCENES, make_nag

    from superpoint_transformer_amd import precision
    precision.set_matrix_precision(args.dtype)
    from superpoint_transformer_amd import csr as _csr
    _csr.use_sub_views(not args.rebuild_csr)
    mix = [SCENE_MIX[r % len(SCENE_MIX)] if (args.scene_mix and world > 1) else 1.0 for r in range(world)]
    if args.mode == "iteration":
        from superpoint_transformer_amd.synthetic import make_raw_nag
        nag = make_raw_nag(args.scene, seed=1234 + rank, device=dev)
    else:
        nag = make_nag(args.scene, seed=1234 + rank, device=dev, graph=args.graph, order=args.order,
                       scale=mix[rank])
    path = hotpath.build(nag, dev, world=world, stages=args.stages, mode=args.mode, model=args.model,
                         kernel_timers=True)
    if args.graph == "random" and args.order == "storage":
        # PMC captures (profiles/traffic.json) exist per (scene, net) of the default generator only
        # (keyed by dtype too: a bf16 line never prints traffic fractions off the f32 capture's bytes)
        path.workload = (f"{args.scene}/{args.model}" + ("" if args.mode == "train" else f"/{args.mode}")
                         + ("" if args.dtype == "f32" else f"/{args.dtype}"))

    # A process that is the first to touch a box's GPU runs its first seconds ~10 % slow (clock
    # ramp; measured: the same binary 74.6 ms/step in the first process of a fresh box, 67.3 in
    # the fifth): untimed steps until `--settle` seconds have passed, then the W warm-up steps.
    t_settle = time.perf_counter()
    while True:
        go = time.perf_counter() - t_settle < args.settle
        if world > 1:
            # every rank must run the same number of steps (a step holds a collective): stop
            # as soon as ANY rank's time is up
            flag = torch.tensor([1.0 if go else 0.0], device=dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            go = bool(flag.item() > 0)
        if not go:
            break
        path.step()
        torch.cuda.synchronize()
    for _ in range(args.warmup):
        path.step()
    captured = None
    if args.capture:
        if not hasattr(path, "capture") or args.mode != "train":
            raise SystemExit("bench.py: --capture applies to --mode train")
        path.capture()
        captured = ("forward + loss + backward + AdamW in one graph" if path._graph_opt else
                    "forward + loss + backward in one graph; all-reduce + AdamW eager")
        for _ in range(2):
            path.step()
    path.reset_kernel_timers()

    device_index = local              # (`local` is reused for the local-graph companion below)

    def barrier():
        if world > 1 or force_dist:
            if share:
                dist.barrier()
            else:
                dist.barrier(device_ids=[device_index])
        torch.cuda.synchronize()

    _log("warm-up done, timing")
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        path.step()
    barrier()
    dt = time.perf_counter() - t0
    _log(f"timed region done: {dt / args.steps * 1e3:.2f} ms per step")
    from superpoint_transformer_amd import parallel
    dt = parallel.max_over_ranks(dt, dev)
    # the adopted views' verdicts that are still on their way (a training loop's last batch:
    # csr.verify_adopted) - outside the timed region, raises on a stale `sub`
    _csr.verify_adopted(block=True)

    n0 = nag.num_points[0]
    # iteration mode: what the chain left for the model to step on (raw points are `value`'s unit)
    path_sizes = list(getattr(path, "last_sizes", None) or []) if args.mode == "iteration" else None
    # every rank's point count is host knowledge (same generator, same scale table)
    n0_all = [max(int(SCENES[args.scene][0] * m), 8) if m != 1.0 else SCENES.get(args.scene, (n0,))[0]
              for m in mix] if args.scene in SCENES else [n0] * world
    n0_all[rank] = n0
    value = sum(n0_all) * args.steps / dt / 1e6
    roof = path.roofline(HBM_PEAK_GBS)
    north_needed = roof.get("achieved") is None      # the step did not launch the stand-alone pool
    workload = (path.describe(args.scene, SCENES.get(args.scene), args.graph)
                if args.stages == "all" else path.describe(args.scene, SCENES.get(args.scene)))
    if args.order == "morton":
        workload += "; level-1/2 nodes stored along a Morton curve"
    elif args.order == "grouped":
        workload += "; nodes in the layout of transforms.MortonOrder (sorted by parent, then Morton curve)"

    # what the step's one collective cost on this run's process group (RCCL over xGMI, or one
    # rank under SPT_FORCE_COLLECTIVES=1); null without a process group
    collective = None
    bucket = getattr(path, "bucket", None)
    if bucket is not None and dist.is_initialized():
        bucket.pack()
        us = bucket.time_allreduce_us()
        collective = {"backend": dist.get_backend(), "ranks": dist.get_world_size(),
                      "bytes": int(bucket.flat.numel() * 4),
                      "allreduce_us": round(us, 1) if us is not None else None}

    # The default "f32" mode runs the attention GEMMs and the fused layers' backward GEMMs as three
    # bf16 products per f32 product (~2^-17 relative per product: 17 of f32's 24 bits; every f32 parity bar holds): print what the
    # same step costs on the f32 matrix pipe everywhere next to it (untimed for `value`).
    exact_ms = None
    if args.dtype == "f32" and world == 1 and args.mode == "train" and not args.no_f32_exact:
        precision.set_matrix_precision("f32-exact")
        for _ in range(2):
            path.step()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(3):
            path.step()
        torch.cuda.synchronize()
        exact_ms = (time.perf_counter() - t1) / 3 * 1e3
        precision.set_matrix_precision(args.dtype)
    # ... and under the reference's other trainer setting (`precision: bf16`, BASELINE config #2's
    # wording): bf16 matrix operands + bf16 storage of the point MLP's layer outputs
    bf16_ms = None
    if args.dtype == "f32" and world == 1 and args.mode == "train" and not args.no_f32_exact:
        precision.set_matrix_precision("bf16")
        for _ in range(2):
            path.step()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(5):
            path.step()
        torch.cuda.synchronize()
        bf16_ms = (time.perf_counter() - t1) / 5 * 1e3
        precision.set_matrix_precision(args.dtype)

    cpu = None
    pre = None
    headline = (args.mode == "train" and args.model == "spt64" and args.stages == "all"
                and args.dtype == "f32")
    # The headline graph is the no-locality stress case BY CHOICE.  Real superpoint graphs are
    # spatially local (src/transforms/graph.py:193-321 builds them by radius search): the same step
    # on the kNN-on-centroids graph with the level-1/2 nodes stored along a Morton curve, next to it.
    local = None
    if (headline and world == 1 and not args.no_local and not args.no_f32_exact
            and args.graph == "random" and args.order == "storage"):
        north = path.northstar(HBM_PEAK_GBS) if north_needed else None   # (before the second scene's memory)
        if north is not None:
            roof.update(north)
        _log("stand-alone north-star kernel timed; building the local-graph scene")
        nag_l = make_nag(args.scene, seed=1234 + rank, device=dev, graph="local", order="grouped")
        path_l = hotpath.build(nag_l, dev, world=world, stages=args.stages, mode=args.mode,
                               model=args.model, kernel_timers=True)
        for _ in range(2):
            path_l.step()
        path_l.reset_kernel_timers()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(5):
            path_l.step()
        torch.cuda.synchronize()
        roof_l = path_l.roofline(HBM_PEAK_GBS)
        local = {"ms_per_step": round((time.perf_counter() - t1) / 5 * 1e3, 4),
                 "workload": path_l.describe(args.scene, SCENES.get(args.scene), "local")
                 + "; nodes in the layout of the load-time transform transforms.MortonOrder (level 2 along a "
                 "Morton curve, level 1 by parent then curve, level-0 points grouped by superpoint)",
                 "kernels": [{k: v for k, v in kk.items() if k in
                              ("kernel", "ms_per_launch", "launches_per_step", "bytes_per_launch",
                               "achieved", "frac")}
                             for kk in roof_l.get("kernels", []) if "attention" in kk["kernel"]]}
        del path_l, nag_l
        _log(f"local-graph steps done: {local['ms_per_step']} ms per step")
    elif north_needed and rank == 0:
        # (N > 1 too: rank 0 times the stand-alone kernel on its own scene - no collective involved -
        # so that every line of a scaling run carries `roofline.achieved`)
        north = path.northstar(HBM_PEAK_GBS)
        if north is not None:
            roof.update(north)
    train_batch = None
    if headline and world == 1 and rank == 0 and not args.no_train_batch and args.scene == "S":
        del path
        path = None
        torch.cuda.empty_cache()
        train_batch = train_batch_leg(dev)
        _log(f"train-batch companion done: {train_batch.get('ms_per_step_T')} ms captured, "
             f"{train_batch.get('ms_per_step_T_eager')} eager, iteration {train_batch.get('ms_per_iteration_T')}")
    if rank == 0 and world == 1 and not headline:
        del path                      # other BASELINE configs: the GPU line only
    elif rank == 0 and world == 1:
        if path is not None:
            del path
        torch.cuda.empty_cache()
        if not args.no_preprocess:
            pre = preprocess_leg(args.scene, n0, dev)
            _log("preprocess leg done")
        if not args.no_cpu_baseline:
            cpu = cpu_baseline(args.scene, args.cpu_scale, all_cores=args.cpu_all_cores)
            _log("cpu baseline done")
            if pre is not None:
                # the SAME cloud size as the GPU leg (a few seconds on the box's cores)
                pre["cpu_baseline"] = cpu_preprocess_baseline(args.scene, n0)
                _log("cpu preprocess baseline done")

    if rank == 0:
        line = {
            "metric": {"infer": "Mpoints/s SPT forward (inference) on a DALES-scale NAG",
                       "iteration": "Mpoints/s of RAW stored points per training iteration (on-device transform "
                                    "chain + SPT fwd+bwd); the model steps on the SAMPLED subset: see "
                                    "value_sampled_points"}.get(args.mode, "Mpoints/s SPT fwd+bwd on S3DIS-scale NAG"),
            "value": round(value, 3),
            "value_sampled_points": (round(int(path_sizes[0]) * args.steps / dt / 1e6 * world, 3)
                                     if path_sizes else None),
            "unit": "Mpoints/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": {"f32": "f32 (the reference's shipped matmul class - configs/train.yaml:60-61 "
                             "`float32_matmul_precision: high` = TF32 / bf16x3 next to `precision: 32` - as "
                             "bf16x3: matrix products on the bf16 pipe with split operands, attention and the "
                             "backward of the fused layers and of the attention blocks' Linears hi*hi + lo*hi + "
                             "hi*lo, the forward of the fused layers and of those Linears the f32-exact 3-way "
                             "split with 6 products; f32 accumulate, storage and statistics)",
                      "f32-exact": "f32 (f32 matrix pipe: the reference under `float32_matmul_precision: highest`)",
                      "bf16": ("bf16 (matrix operands and the point MLP's stored layer outputs; f32 "
                               "accumulate, statistics, gradients, segment / attention tensors)"
                               if precision.bf16_activation_storage() else
                               "bf16 (matrix operands; f32 accumulate, storage and statistics)")}[args.dtype],
            "ms_per_step_f32_exact": round(exact_ms, 4) if exact_ms else None,
            "ms_per_step_bf16": round(bf16_ms, 4) if bf16_ms else None,
            "ms_per_step_local": local["ms_per_step"] if local else None,
            "local_graph": local,
            "captured_graph": captured,
            "ms_per_step_T": train_batch.get("ms_per_step_T") if train_batch else None,
            "ms_per_iteration_T": train_batch.get("ms_per_iteration_T") if train_batch else None,
            "train_batch": train_batch,
            "data": "synthetic",
            "config": {
                "workload": workload,
                "scene": args.scene,
                "points_per_gpu": n0 if len(set(n0_all)) == 1 else n0_all,
                "scene_mix": list(mix) if args.scene_mix and world > 1 else None,
                "parallelism": f"dp{world}",
                "mode": args.mode,
                "net": args.model,
                "graph": args.graph,
                "csr_views": "rebuilt" if args.rebuild_csr else "nag.sub",
            },
            "roofline": roof,
            "collective": collective,
            "cpu_baseline": cpu,
            "preprocess": pre,
        }
        print(json.dumps(line))
    if world > 1 or force_dist:
        barrier()                     # the other ranks leave with rank 0, not while it still measures
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
