"""The benchmarked hot path: one step = one pass over one NAG batch.

``SPTTrainStep`` (default): SPT-64 / spt-2 of the S3DIS experiment
(configs/experiment/semantic/s3dis.yaml:6-9 -> configs/model/semantic/spt-2.yaml)
forward + multi-level cross-entropy + backward + AdamW step on one synthetic
NAG batch resident in HBM, including the per-batch CSR builds (device sorts of
every ``super_index`` / ``edge_index[0]``).  N > 1: every rank owns its own
scene; the ~0.85 MB of gradients meet in ONE flat all-reduce per step over
RCCL (``parallel.FlatGradAllReduce``).

``ScatterChain``: only the hierarchical segment-CSR scatter kernels (max pool
fwd/bwd L0->L1->L2 + unpool fwd/bwd), kept as a kernel-level benchmark.
"""
import os

import torch
from torch import nn

from . import _lib
from . import csr as _csr
from . import ops, parallel
from .nn import SPT, Classifier, GraphNorm

NUM_CLASSES = 13  # S3DIS


def spt64_config(point_in=8, edge_in=18):
    """Widths derived from the reference's config tree: point MLP
    [12,32,64,128]; h_edge_mlp [18,32,32]; down in_mlps [132,64,64], [68,64,64];
    up in_mlp [132,64,64]; 16 heads, qk_dim 4, 3 blocks down / 1 up, no FFN."""
    inj = 3 + 1  # normalised position + parent diameter
    return dict(
        point_mlp=[point_in + inj, 32, 64, 128], down_dim=[64, 64], down_pool_dim=[128, 64],
        down_in_mlp=[[inj + 128, 64, 64], [inj + 64, 64, 64]], down_num_heads=16,
        down_num_blocks=3, down_ffn_ratio=1, up_dim=[64], up_in_mlp=[[inj + 64 + 64, 64, 64]],
        up_num_heads=16, up_num_blocks=1, up_ffn_ratio=1, node_mlp=None,
        h_edge_mlp=[edge_in, 32, 32], mlp_norm=GraphNorm, norm=GraphNorm, qk_dim=4,
        in_rpe_dim=32, k_rpe=True, q_rpe=True, v_rpe=True, no_ffn=True, use_pos=True,
        use_node_hf=True, use_diameter_parent=True, pool="max", unpool="index",
        fusion="cat", output_stage_wise=True)


def nano2_config(seg_in=8, edge_in=18):
    """configs/model/semantic/nano-2.yaml on datamodule/semantic/s3dis_nano.yaml: no level-0
    stage, the NAG starts at level 1 with 8 handcrafted segment features through a node MLP to
    16, dims 16, qk_dim 2 (value dim 1), edge MLP to 16."""
    cfg = spt64_config(8, edge_in)
    d, inj = 16, 3 + 1 + 16
    cfg.update(nano=True, point_mlp=None, down_dim=[d, d], down_pool_dim=[128, d],
               down_in_mlp=[[inj, d, d], [inj + d, d, d]], up_dim=[d],
               up_in_mlp=[[inj + d + d, d, d]], node_mlp=[seg_in, d, d],
               h_edge_mlp=[edge_in, d, d], qk_dim=2, in_rpe_dim=d)
    return cfg


def spt128_config(point_in=8, edge_in=18):
    """SPT-128 of configs/experiment/semantic/kitti360.yaml:22-27: the same spt-2 tree
    (2 down stages, 1 up) with `_down_dim` / `_up_dim` 128 (value dim 8 per head),
    `no_ffn: False`, `down_ffn_ratio: 1`."""
    cfg = spt64_config(point_in, edge_in)
    inj = 3 + 1
    cfg.update(down_dim=[128, 128], down_pool_dim=[128, 128],
               down_in_mlp=[[inj + 128, 128, 128], [inj + 128, 128, 128]],
               up_dim=[128], up_in_mlp=[[inj + 128 + 128, 128, 128]], no_ffn=False)
    return cfg


def _pmc_traffic(op, workload):
    """HBM bytes per launch of a roofline op as measured with rocprofv3 PMC counters on THIS
    workload (profiles/traffic.json, written by tools/make_traffic_json.py from a
    tools/pmc_step.sh capture; keys ``<op>@<scene>/<net>``; FETCH_SIZE doubled per
    MI355X_MICROARCH.md's gfx950 correction).  None when no capture exists for the workload -
    a line of another scene or model never borrows scene S's bytes."""
    import json
    import os
    if not workload:
        return None
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                        "profiles", "traffic.json")
    try:
        with open(path) as f:
            return json.load(f).get(f"{op}@{workload}") or None
    except (OSError, ValueError):
        return None


def _kernel_names(mode):
    """Kernels behind the four timed ops in matrix mode ``mode`` (precision.get_matrix_precision):
    what the dispatch of csrc/edge_attn.hip / fused_mlp.hip launches for the SPT-64 head layout."""
    if mode == "f32-exact":
        return {"attn_bwd": "hipMemsetAsync + spt::mfma::attn_bwd_packed_kernel<0> (f32 matrix pipe) + "
                            "spt::attn_reduce_partials_kernel",
                "attn_fwd": "spt::mfma::attn_fwd_mfma_kernel<0> (f32 matrix pipe)",
                "mlp_bwd_pooled": None,        # mode 0 has no pooled backward: dense route
                "mlp_fwd": "spt::fmlp::fwd_kernel<16, 8> (f32 matrix pipe)"}
    p = 3 if mode == "f32" else 1
    lo = "true" if mode == "f32" else "false"
    from . import precision
    st = mode == "bf16" and precision.bf16_activation_storage()   # layer outputs stored as bf16
    to = bool(_lib.lib.spt_attn_bwd_el_target_order(-1))           # edge stream by target / by source
    return {"attn_bwd": (f"spt::to::attn_bwd_to_prep_kernel + spt::to::attn_bwd_to_kernel<{p}> + "
                         "spt::to::attn_q_reduce_kernel + spt::attn_reduce_partials_kernel" if to else
                         f"spt::el::attn_bwd_prep_kernel + spt::el::attn_bwd_el_kernel<{p}> + "
                         "spt::el::attn_kv_reduce_kernel + spt::attn_reduce_partials_kernel"),
            "attn_fwd": f"spt::mfma::attn_fwd_mfma_kernel<{p}>",
            "mlp_bwd_pooled": f"spt::fdma::bwd_dma_kernel<64, 128, 8, 2, {lo}, true" + (", true>" if st else ">"),
            "mlp_fwd_pool": (f"spt::fpool::fwd_pool_kernel<64, 128, {3 if mode == 'f32' else 1}, "
                             f"{'true' if st else 'false'}> + gram_tables_kernel + pool_apply_kernel"),
            "mlp_bwd_pool": (f"spt::fpool::pool_bwd_gm_kernel + pool_bwd_coef_kernel + "
                             f"bwd_pool_kernel<64, 128, {lo}, {'true' if st else 'false'}, 8> + "
                             "reduce_tables + pool_bwd_gw_dense_kernel"),
            "mlp_fwd": ("spt::fmlp::fwd_kernel_x3<16, 8> (f32 product as 6 bf16 products of 3-way split operands)"
                        if _lib.lib.spt_fused_linear_fwd_use_x3(-1) else "spt::fmlp::fwd_kernel<16, 8>")
                       if mode == "f32"
                       else ("spt::fmlp::fwd_kernel_bf<16, 8, false, true, true>" if st
                             else "spt::fmlp::fwd_kernel_bf<16, 8, false>")}


class SPTSegmenter(nn.Module):
    """SPT backbone + one Classifier per output level (semantic.py:291-294)."""

    def __init__(self, **cfg):
        super().__init__()
        self.net = SPT(**cfg)
        dims = self.net.out_dim if isinstance(self.net.out_dim, list) else [self.net.out_dim]
        self.head = nn.ModuleList([Classifier(d, NUM_CLASSES) for d in dims])

    def forward(self, nag):
        outs = self.net(nag)
        outs = outs if isinstance(outs, list) else [outs]
        return [h(x) for h, x in zip(self.head, outs)]


class SPTPanoptic(nn.Module):
    """PanopticSegmentationModule.forward up to the partitioner (src/models/panoptic.py:443-492):
    SPT backbone, one Classifier per output level, and the edge-affinity head
    ``MLP([2 * up_dim, 32, 16, 1], norm=None, last_activation=False)`` of
    configs/model/panoptic/_instance.yaml:21-28 on the symmetric edge features of
    ``obj_edge_index``.  The point encoder is the panoptic configs' smaller one
    (``_point_mlp: [32, 64, 64]``, _instance.yaml:16)."""

    def __init__(self, **cfg):
        super().__init__()
        from .nn import MLP
        self.net = SPT(**cfg)
        dims = self.net.out_dim if isinstance(self.net.out_dim, list) else [self.net.out_dim]
        self.head = nn.ModuleList([Classifier(d, NUM_CLASSES) for d in dims])
        self.edge_affinity_head = MLP([2 * dims[0], 32, 16, 1], activation=nn.LeakyReLU(),
                                      norm=None, last_norm=False, last_activation=False)

    def forward(self, nag):
        outs = self.net(nag)
        outs = outs if isinstance(outs, list) else [outs]
        logits = [h(x) for h, x in zip(self.head, outs)]
        x_edge = ops.edge_affinity_features(outs[0], nag[1]["obj_edge_index"])   # panoptic.py:477-480
        return logits, self.edge_affinity_head(x_edge).squeeze(-1)              # :481-483


def panoptic_config(point_in=8, edge_in=18):
    """spt-2 of configs/model/panoptic/spt-2.yaml: the semantic tree with the point encoder
    of _instance.yaml:16 ([32, 64, 64]: pooled width 64)."""
    cfg = spt64_config(point_in, edge_in)
    inj = 3 + 1
    cfg.update(point_mlp=[point_in + inj, 32, 64, 64], down_pool_dim=[64, 64],
               down_in_mlp=[[inj + 64, 64, 64], [inj + 64, 64, 64]])
    return cfg


MODEL_CONFIGS = {"spt64": spt64_config, "spt128": spt128_config}


class _NagView:
    """Minimal NAG-like view: ``nag[i]`` -> level dict, ``nag.num_clouds``."""

    def __init__(self, nag):
        self.levels = nag.levels
        self.num_clouds = nag.num_clouds

    def __getitem__(self, i):
        return self.levels[i]


class SPTTrainStep:
    name = "SPT-64 (spt-2, S3DIS cfg) fwd + CE loss + bwd + AdamW, incl. per-batch CSR builds"

    def __init__(self, nag, dev, world=1, seed=0, model="spt64", kernel_timers=False):
        """``kernel_timers``: HIP-event timers around the north-star kernel and the ops leading the
        step (what ``roofline()`` reports) - a measurement aid, off unless asked for (bench.py)."""
        self.nag, self.dev, self.world = _NagView(nag), dev, world
        self.n = nag.num_points
        self.net_name = model
        self.workload = None                    # "<scene>/<net>": set by the caller that knows the scene
        torch.manual_seed(seed)
        if model != "spt64":
            self.name = self.name.replace("SPT-64 (spt-2, S3DIS cfg)",
                                          "SPT-128 (spt-2 tree, KITTI-360 cfg: dim 128, FFN on)")
        self.model = SPTSegmenter(**MODEL_CONFIGS[model](nag[0]["x"].shape[1],
                                                         nag[1]["edge_attr"].shape[1])).to(dev)
        self.params = [p for p in self.model.parameters()]
        parallel.broadcast_parameters(self.params, src=0)
        self.bucket = parallel.FlatGradAllReduce(
            self.params, always=os.environ.get("SPT_FORCE_COLLECTIVES") == "1")
        # one multi-tensor kernel for the 130 parameter tensors (matters at train-batch sizes,
        # where the step is launch-bound)
        self.opt = torch.optim.AdamW(self.params, lr=1e-3, weight_decay=1e-4,
                                     fused=dev.type == "cuda")
        g = torch.Generator(device=dev).manual_seed(5)
        self.labels = [torch.randint(0, NUM_CLASSES, (self.n[i],), device=dev, generator=g)
                       for i in (1, 2)]
        self.lambdas = [1.0, 50.0]               # configs/model/semantic/default.yaml:12
        self.loss_fn = ops.cross_entropy          # CrossEntropyLoss(), mean reduction, on csrc/loss.hip
        n0, c = self.n[0], 128
        self.tname = f"segcsr_reduce_fwd:3:{n0}x{c}"
        self.k_timers = None
        if kernel_timers:
            ops.enable_timer(self.tname)
            self._enable_kernel_timers()
        self.last_loss = None

    def _enable_kernel_timers(self):
        """HIP-event timers (launch stream) around the ops whose kernels lead the step's GPU time."""
        n1 = self.n[1]
        e1 = self.nag.levels[1]["edge_index"].shape[1]
        self.k_timers = {
            "attn_bwd": f"edge_attn_bwd:{n1}:{e1}",
            "attn_fwd": f"edge_attn_fwd:{n1}:{e1}",
            "mlp_bwd_pooled": "fused_linear_bwd_pooled:64x128:",
            "mlp_fwd": "fused_linear_fwd:64x128:",
            # round 5: the top layer with the pool inside (no [rows, 128] tensor)
            "mlp_fwd_pool": "fused_linear_fwd_pool:64x128:",
            "mlp_bwd_pool": "fused_linear_bwd_pool:64x128:",
        }
        for key, name in self.k_timers.items():
            ops.enable_timer(name, prefix=key.startswith("mlp"))
        self.level1 = (n1, e1)

    def reset_kernel_timers(self):
        ops.reset_timers()
        self._timed_steps = 0

    def _forget_csr(self):
        for lv in self.nag.levels:
            _csr.forget(lv.get("super_index"), lv.get("edge_index"), lv.get("batch"))

    _timed_steps = 0
    graph = None                      # a captured step (capture()): replayed by step()

    def _fwd_bwd(self):
        """Forward + loss + backward of the batch: every CSR view, run table and gradient rebuilt."""
        self._forget_csr()
        logits = self.model(self.nag)
        loss = sum(l * self.loss_fn(lg, y) for l, lg, y in zip(self.lambdas, logits, self.labels))
        self.bucket.zero()
        loss.backward()
        return loss

    def step(self):
        self._timed_steps += 1
        if self.graph is not None:
            return self._replay()
        loss = self._fwd_bwd()
        self.bucket.reduce()          # one RCCL all-reduce of the flat gradient (no-op at N=1)
        self.opt.step()
        self.last_loss = loss
        return loss

    def capture(self, warmup=3):
        """Capture the step of THIS batch into one graph (hipGraph through ``torch.cuda.CUDAGraph``)
        and make ``step()`` replay it: the train-batch regime runs ~370 launches of a few
        microseconds each (reference batches are capped to fixed sizes,
        configs/datamodule/semantic/default.yaml:79-80 ``max_num_nodes`` / ``max_num_edges``), so
        the host's ~370 ctypes / autograd round trips per step and their jitter - what a DDP rank
        shows its peers as skew in front of the all-reduce - leave the step: one graph launch.

        What the graph holds: the per-batch CSR builds and adopted-view checks (their verdict
        stays a device flag, read by ``csr.verify_adopted(block=True)`` OUTSIDE the graph), the
        forward, the losses, the backward and - with one rank - the AdamW update.  With more than
        one rank (or ``SPT_FORCE_COLLECTIVES=1``) the graph ends after the backward; the flat
        all-reduce and the fused AdamW (6 launches) follow eagerly, so that no collective is
        captured.  The batch is static by construction: a new batch of the same caps must be
        COPIED into the captured tensors (``nag`` levels, ``labels``), host-side run tables are
        those of the captured batch.  Kernel timers are off in a captured step (no timing events
        inside a capture).  Returns self."""
        from . import ops as _ops
        from . import csr as _csr_mod
        if self.graph is not None:
            return self
        dev = self.dev
        _csr_mod.verify_adopted(block=True)          # nothing pending from eager steps
        # An autograd graph of an EAGER step that is still alive (through `last_loss`) keeps the
        # parameters' AccumulateGrad nodes, which remember the stream they were created on (the
        # default stream): the backward of the capture would then hop to that stream - outside the
        # capture.  Drop it; the warm-up below creates the nodes on the capture's own stream.
        self.last_loss = None
        import gc
        gc.collect()
        collective = self.bucket.world > 1 or (self.bucket.always and
                                               torch.distributed.is_initialized())
        self._graph_opt = not collective
        if self._graph_opt:
            for g in self.opt.param_groups:            # the device-side step counter is advanced in-graph
                g["capturable"] = True
            if not self.opt.state:
                warmup = max(warmup, 1)                # AdamW's moments must exist BEFORE the capture
        paused = _ops.pause_timers(True)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        try:
            with torch.cuda.stream(side):
                for _ in range(warmup):                # allocator, workspaces, optimizer state
                    self._fwd_bwd()
                    if self._graph_opt:
                        self.opt.step()
            torch.cuda.current_stream(dev).wait_stream(side)
            torch.cuda.synchronize(dev)
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph, stream=side, capture_error_mode="thread_local"):
                loss = self._fwd_bwd()
                if self._graph_opt:
                    self.opt.step()
            self._graph_loss = loss
            # the gradient tensors autograd allocated inside the capture: the replay writes THESE
            self._graph_grads = [p.grad for p in self.params]
        finally:
            _ops.pause_timers(paused)
        torch.cuda.synchronize(dev)
        return self

    def _replay(self):
        self.graph.replay()
        if not self._graph_opt:
            for p, g in zip(self.params, self._graph_grads):
                p.grad = g                             # (pack() re-pointed them at the flat buffer)
            self.bucket._packed = False
            self.bucket.reduce()
            self.opt.step()
        self.last_loss = self._graph_loss
        return self._graph_loss

    def roofline(self, peak_gbs):
        """North-star entry (the L0->L1 segment-CSR max) + `kernels`: the ops leading the step's
        GPU time, each with its SURVEY 8(d) algorithmic bytes / FLOPs per launch and the mean
        duration measured with HIP events on the launch stream over the timed steps.  Kernel
        names follow the matrix mode that ran; bytes per launch are the op's bytes per step over
        its launches per step (a batch of B clouds may take B launches); PMC traffic only where
        a capture of this workload exists (profiles/traffic.json)."""
        from . import precision
        n0, n1 = self.n[0], self.n[1]
        c = getattr(self, "pool_c", 128)
        st16 = (precision.get_matrix_precision() == "bf16" and precision.bf16_activation_storage()
                and c == 128 and n0 >= 65536)
        bytes_ = n0 * ((2 if st16 else 4) * c + 4) + n1 * (8 * c + 4)
        ms = ops.timer_mean_ms(self.tname)
        ach = bytes_ / (ms * 1e-3) / 1e9 if ms else None
        traffic = _pmc_traffic("segmax", self.workload)
        roof = {"bound": "hbm",
                "kernel": ("spt::segmax_stream_kernel<true, true> (bf16 rows; " if st16 else
                           "spt::segmax_stream_kernel<true> (") +
                          "L0->L1 segment max + arg, C=%d, the point MLP's "
                          "last GraphNorm + LeakyReLU applied on the fly)" % c
                if c == 128 and n0 >= 65536 else "spt::segcsr_reduce_kernel<3, 4, true> (L0->L1 segment max + arg)",
                "achieved": round(ach, 1) if ach else None, "peak": peak_gbs,
                "unit": "GB/s", "frac": round(ach / peak_gbs, 4) if ach else None,
                "traffic": traffic.get("bytes") if traffic else None,
                "traffic_source": traffic.get("source") if traffic else None,
                "bytes_per_launch": bytes_,
                "ms_per_launch": round(ms, 4) if ms else None}
        kernels = []
        if getattr(self, "k_timers", None):
            n1, e1 = self.level1
            names = _kernel_names(precision.get_matrix_precision())
            ab = 2 if (precision.get_matrix_precision() == "bf16"
                       and precision.bf16_activation_storage()) else 4
            rows = n0
            steps = max(self._timed_steps, 1)
            # bytes / FLOPs of the op PER STEP (all launches of the op in one step together):
            # backward 264 B/edge + 2 176 B/node, ~37.4 kFLOP/edge (3 GEMMs of the
            # forward's 12.3 kFLOP + the per-edge math); forward 136 B/edge + 1 156 B/node, 12.6 kFLOP;
            # fused tall-MLP layers, 64 -> 128 at level 0: forward reads x (4K) writes h (4N) per row;
            # pooled backward reads h (4N) + x (4K), writes gx (4K) per row, + (gout, arg) per segment
            spec = {
                "attn_bwd": ("one level-1 attention backward: all kernels of the call",
                             264 * e1 + 2176 * n1, 37.4e3 * e1, True),
                "attn_fwd": ("one level-1 attention forward", 136 * e1 + 1156 * n1, 12.6e3 * e1, True),
                # (ab = bytes per stored activation value: 2 with the bf16 mode's storage option)
                "mlp_bwd_pooled": ("64 -> 128 backward of the point MLP's top layer with the L0->L1 "
                                   "pool's backward inside",
                                   rows * (ab * 128 + ab * 64 + 4 * 64) + n1 * 1024,
                                   2 * 2 * 64 * 128 * rows, False),
                "mlp_fwd": ("64 -> 128 forward of the point MLP's top layer",
                            rows * ab * (64 + 128), 2 * 64 * 128 * rows, False),
                # pool-fused top layer (csrc/fused_pool.hip): forward reads x (ab K) + the row id and
                # the segment id (8) per row, writes (raw, arg) and reads / writes (raw, out) per
                # (segment, channel): 5 x 4 N per segment; backward reads x (ab K) + ids (8), writes
                # gx (4 K) per row, and moves (gout, raw | gm | gm, arg) = 5 x 4 N per segment
                "mlp_fwd_pool": ("64 -> 128 forward of the point MLP's top layer WITH the L0->L1 max-pool, "
                                 "the norm's statistics from the input's Gram matrix: product + pool + "
                                 "tables + apply, all kernels of the call",
                                 rows * (ab * 64 + 8) + n1 * 5 * 4 * 128,
                                 2 * (64 * 128 + 64 * 64 // 2) * rows, False),
                "mlp_bwd_pool": ("64 -> 128 backward of the same unit from (gout, arg, raw): gm + "
                                 "coefficients + main kernel + reductions, all kernels of the call",
                                 rows * (ab * 64 + 8 + 4 * 64) + n1 * 5 * 4 * 128,
                                 2 * (2 * 64 * 128 + 64 * 64) * rows, False),
            }
            for key, (what, kbytes, kflops, per_call) in spec.items():
                tms = ops.timer_mean_ms(self.k_timers[key])
                if not tms or names.get(key) is None:
                    continue
                per_step = ops.timer_count(self.k_timers[key]) / steps
                if not per_call and per_step > 1:
                    # the layer runs as one launch per cloud: the figures above cover the layer
                    kbytes, kflops = kbytes / per_step, kflops / per_step
                gbs = kbytes / (tms * 1e-3) / 1e9
                tfs = kflops / (tms * 1e-3) / 1e12
                tr = _pmc_traffic(key, self.workload)
                kernels.append({
                    "kernel": f"{names[key]} ({what})", "ms_per_launch": round(tms, 4),
                    "launches_per_step": round(per_step, 2),
                    "bytes_per_launch": int(kbytes), "achieved": round(gbs, 1), "unit": "GB/s",
                    "frac": round(gbs / peak_gbs, 4),
                    "flops_per_launch": float(kflops), "achieved_tflops_f32_equivalent": round(tfs, 2),
                    "traffic": tr.get("bytes") if tr else None,
                    # what the memory system actually moved per launch (PMC) over the same time:
                    # the kernels of this list gather / scatter rows, their traffic is a multiple
                    # of the algorithmic bytes by construction (k / v rows per EDGE, not per node)
                    "traffic_gbs": round(tr["bytes"] / (tms * 1e-3) / 1e9, 1) if tr else None,
                    "traffic_frac": round(tr["bytes"] / (tms * 1e-3) / 1e9 / peak_gbs, 4) if tr else None})
        roof["kernels"] = kernels
        return roof

    def northstar(self, peak_gbs, reps=20):
        """The north-star segment-CSR scatter kernel on its own: ``MaxPool.__call__`` of the
        level-0 -> level-1 pool (src/nn/pool.py:61-82) = ``ops.segment_reduce(x, super_index, max,
        arg)`` on a [N0, 128] f32 tensor through the scene's own CSR view, timed with HIP events on
        the launch stream.  Since round 5 the TRAINING STEP no longer launches this kernel (the
        pool runs inside the top layer's product, csrc/fused_pool.hip): the operator stays on the
        boundary (nn.MaxPool, shims.scatter_shim.scatter_max) and is measured here, in the same
        process, on the same index.  Returns the fields of the bench line's ``roofline`` object,
        (the caller asks when the step's own timer of this kernel stayed empty)."""
        n0, n1 = self.n[0], self.n[1]
        c = 128
        si = self.nag.levels[0].get("super_index")
        if si is None or n0 < 65536:
            return None
        x = torch.randn(n0, c, device=self.dev)
        for _ in range(2):
            ops.segment_reduce(x, si, n1, "max", return_arg=True)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            ops.segment_reduce(x, si, n1, "max", return_arg=True)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        del x
        # row 4 C + 4 (its position's row id), parent 4 C (max) + 4 C (arg, int32) + 4 (range)
        bytes_ = n0 * (4 * c + 4) + n1 * (8 * c + 4)
        ach = bytes_ / (ms * 1e-3) / 1e9
        tr = _pmc_traffic("segmax_standalone", self.workload)
        return {"bound": "hbm",
                "kernel": "spt::segmax_stream_kernel<false> (L0->L1 segment max + arg of [N0, 128] f32 rows "
                          "through the level's CSR view: nn.MaxPool / scatter_max on its own - the step "
                          "runs the pool inside spt::fpool::fwd_pool_kernel, see `kernels`)",
                "achieved": round(ach, 1), "peak": peak_gbs, "unit": "GB/s",
                "frac": round(ach / peak_gbs, 4),
                "traffic": tr.get("bytes") if tr else None,
                "traffic_source": tr.get("source") if tr else None,
                "bytes_per_launch": bytes_, "ms_per_launch": round(ms, 4),
                "measured": f"{reps} stand-alone launches after the timed region, HIP events on the launch stream"}

    def describe(self, scene, sizes, graph="random"):
        mirror = (_csr._USE_MIRROR and getattr(self.nag.levels[1].get("edge_index"), _csr.MIRROR_ATTR, None)
                  is not None)
        edges = ("by-source edge views sorted per step, the by-target stream taken from the lists' mirror "
                 "structure [i<j | j>i | loops] (checked on the device)" if mirror
                 else "edge views (by source, by target) sorted per step")
        views = (f"level CSR views taken from the NAG's stored `sub` (nag[i+1].sub), {edges}"
                 if _csr._USE_SUB_VIEWS and self.nag.levels[1].get("sub") is not None
                 else f"every level CSR view rebuilt by the device sort per step (--rebuild-csr: a NAG without `sub`), {edges}")
        kind = {"random": "uniformly random superpoint graph = no locality, the worst case for the "
                          "attention's k / v gathers (stress case)",
                "local": "kNN-on-centroids superpoint graph (SURVEY 8d), nodes in storage order"}[graph]
        return (f"{self.name}; synthetic NAG scene {scene} (N0,N1,N2,E1,E2,clouds)={sizes}; {kind}; {views}")


class SPTInferStep(SPTTrainStep):
    """Config #3 (DALES tile inference): the forward pass alone under ``no_grad``, model in
    eval mode, per-batch CSR builds included (src/models/semantic.py forward at test time)."""
    name = "SPT-64 (spt-2) forward only (eval, no_grad), incl. per-batch CSR builds"

    def __init__(self, nag, dev, world=1, seed=0, model="spt64", kernel_timers=False):
        super().__init__(nag, dev, world, seed, model, kernel_timers=kernel_timers)
        self.name = SPTInferStep.name if model == "spt64" else SPTInferStep.name.replace("SPT-64", "SPT-128")
        self.model.eval()

    def step(self):
        self._forget_csr()
        self._timed_steps += 1
        with torch.no_grad():
            logits = self.model(self.nag)
        self.last_loss = logits[0]
        return logits


class SPTPanopticStep(SPTTrainStep):
    """Config #5 (SuperCluster panoptic training): backbone + semantic heads + edge-affinity
    head; loss = multi-stage CE + BCE-with-logits on the affinities of ``obj_edge_index``
    (src/models/panoptic.py:443-492, 1100-1160; the graph-cluster partitioner runs at
    validation time only: partition_every_n_epoch, _instance.yaml:33-36)."""
    name = ("SuperCluster panoptic (spt-2 + edge-affinity head) fwd + CE/BCE loss + bwd + AdamW, "
            "incl. per-batch CSR builds")

    def __init__(self, nag, dev, world=1, seed=0, model="spt64", kernel_timers=False):
        self.nag, self.dev, self.world = _NagView(nag), dev, world
        self.n = nag.num_points
        self.net_name, self.workload, self.k_timers = "spt64-panoptic", None, None
        torch.manual_seed(seed)
        self.model = SPTPanoptic(**panoptic_config(nag[0]["x"].shape[1],
                                                   nag[1]["edge_attr"].shape[1])).to(dev)
        # obj_edge_index = the trimmed (i < j) level-1 graph (src/transforms/instance.py:164,199)
        ei = nag[1]["edge_index"]
        self.nag.levels[1]["obj_edge_index"] = ei[:, ei[0] < ei[1]].contiguous()
        self.params = [p for p in self.model.parameters()]
        parallel.broadcast_parameters(self.params, src=0)
        self.bucket = parallel.FlatGradAllReduce(self.params)
        self.opt = torch.optim.AdamW(self.params, lr=1e-3, weight_decay=1e-4,
                                     fused=dev.type == "cuda")
        g = torch.Generator(device=dev).manual_seed(5)
        self.labels = [torch.randint(0, NUM_CLASSES, (self.n[i],), device=dev, generator=g)
                       for i in (1, 2)]
        ne = self.nag.levels[1]["obj_edge_index"].shape[1]
        self.affinity = (torch.rand(ne, device=dev, generator=g) < 0.5).float()
        self.lambdas = [1.0, 50.0]
        self.loss_fn = ops.cross_entropy          # CrossEntropyLoss(), mean reduction, on csrc/loss.hip
        self.bce = nn.BCEWithLogitsLoss()
        self.pool_c = 64
        self.tname = f"segcsr_reduce_fwd:3:{self.n[0]}x64"
        if kernel_timers:
            ops.enable_timer(self.tname)
        self.last_loss = None

    def step(self):
        self._forget_csr()
        self._timed_steps += 1
        logits, aff = self.model(self.nag)
        loss = sum(l * self.loss_fn(lg, y) for l, lg, y in zip(self.lambdas, logits, self.labels))
        loss = loss + self.bce(aff, self.affinity)          # edge_affinity_loss_lambda: 1
        self.bucket.zero()
        loss.backward()
        self.bucket.reduce()
        self.opt.step()
        self.last_loss = loss
        return loss

class ScatterChain:
    name = "segment-CSR scatter chain (CSR build + max-pool fwd/bwd over L0->L1->L2 + unpool fwd/bwd)"
    workload, k_timers, _timed_steps = None, None, 0

    def __init__(self, nag, dev, world=1):
        self.nag, self.dev, self.world = nag, dev, world
        n0, n1, n2 = nag.num_points
        self.n = (n0, n1, n2)
        g = torch.Generator(device=dev).manual_seed(99)
        self.x0 = torch.randn(n0, 128, device=dev, generator=g)
        self.x1 = torch.randn(n1, 64, device=dev, generator=g)
        self.g1 = torch.randn(n1, 128, device=dev, generator=g)
        self.g2 = torch.randn(n2, 64, device=dev, generator=g)
        self.tname = f"segcsr_reduce_fwd:3:{n0}x128"
        ops.enable_timer(self.tname)

    def reset_kernel_timers(self):
        ops.reset_timers()

    def step(self):
        nag = self.nag
        n0, n1, n2 = self.n
        csr0 = _csr.build_csr(nag[0]["super_index"], n1)
        csr1 = _csr.build_csr(nag[1]["super_index"], n2)
        p1, a1 = ops._seg_reduce_fwd(self.x0, csr0, 3, True)
        p2, a2 = ops._seg_reduce_fwd(self.x1, csr1, 3, True)
        u1 = ops._gather_fwd(p2, csr1.idx)
        gp2, _ = ops._seg_reduce_fwd(u1, csr1, 0, False)          # unpool bwd
        gx1 = ops._seg_reduce_bwd(self.g2, a2, csr1, 3, n1)
        gx0 = ops._seg_reduce_bwd(self.g1, a1, csr0, 3, n0)
        return gx0, gx1, gp2

    roofline = SPTTrainStep.roofline

    def describe(self, scene, sizes):
        return f"{self.name}; scene {scene} {sizes}"


class SPTIterationStep(SPTTrainStep):
    """A training ITERATION as the reference runs it (src/datamodules/base.py:341-380:
    ``on_after_batch_transfer`` applies ``on_device_train_transform`` to the batch, then the model
    steps): the per-batch on-device transform chain of configs/datamodule/semantic/default.yaml:
    206-290 - NodeSize -> SampleSubNodes -> SampleSegments -> OnTheFlyHorizontalEdgeFeatures
    (+ self loops) -> SampleEdges - on a RAW NAG as it sits on disk (trimmed edges with the 7
    stored attributes, no node sizes, ``superpoint_transformer_amd.synthetic.make_raw_nag``), then
    forward + CE loss + backward + AdamW on what the chain produced.  Every step starts from a
    fresh copy of the raw batch, so the chain, every CSR view and every run table are rebuilt:
    nothing of the batch is prepared outside the timed region."""
    name = ("one training iteration: on-device transform chain (NodeSize, SampleSubNodes, SampleSegments, "
            "OnTheFlyHorizontalEdgeFeatures, SampleEdges) + SPT-64 fwd + CE loss + bwd + AdamW")

    def __init__(self, raw_nag, dev, world=1, seed=0, model="spt64", kernel_timers=False):
        from . import transforms as T
        self.raw, self.dev, self.world = raw_nag, dev, world
        self.n = raw_nag.num_points
        self.net_name, self.workload, self.k_timers = model, None, None
        self.chain = [T.NodeSize(0), T.SampleSubNodes(1, 0, n_max=32, n_min=8),
                      T.SampleSegments(ratio=0.2), T.OnTheFlyHorizontalEdgeFeatures(),
                      T.SampleEdges(levels=(1, 2), n_min=4, n_max=16)]
        torch.manual_seed(seed)
        probe = self._prepare()
        self.model = SPTSegmenter(**MODEL_CONFIGS[model](probe[0].x.shape[1],
                                                         probe[1].edge_attr.shape[1])).to(dev)
        self.params = [p for p in self.model.parameters()]
        parallel.broadcast_parameters(self.params, src=0)
        self.bucket = parallel.FlatGradAllReduce(
            self.params, always=os.environ.get("SPT_FORCE_COLLECTIVES") == "1")
        self.opt = torch.optim.AdamW(self.params, lr=1e-3, weight_decay=1e-4, fused=dev.type == "cuda")
        self.lambdas = [1.0, 50.0]
        self.loss_fn = ops.cross_entropy
        self.tname = "unused"
        self.last_loss = None
        self.last_sizes = probe.num_points
        self.ms_chain = []

    def _prepare(self):
        nag = self.raw.clone()
        for t in self.chain:
            nag = t(nag)
        return nag

    def step(self):
        self._timed_steps += 1
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        nag = self._prepare()
        e1.record()
        self.ms_chain.append((e0, e1))
        n = nag.num_points
        # labels of the sampled nodes (the reference carries them through the selection; random
        # here: synthetic data has none)
        labels = [torch.randint(0, NUM_CLASSES, (n[i],), device=self.dev) for i in (1, 2)]
        logits = self.model(nag)
        loss = sum(l * self.loss_fn(lg, y) for l, lg, y in zip(self.lambdas, logits, labels))
        self.bucket.zero()
        loss.backward()
        self.bucket.reduce()
        self.opt.step()
        self.last_loss, self.last_sizes = loss, n
        return loss

    def reset_kernel_timers(self):
        self.ms_chain = []
        self._timed_steps = 0

    def roofline(self, peak_gbs):
        torch.cuda.synchronize()
        ms = [a.elapsed_time(b) for a, b in self.ms_chain]
        return {"bound": "hbm", "kernel": "n/a (iteration line: see ms_transform_chain)", "achieved": None,
                "peak": peak_gbs, "unit": "GB/s", "frac": None, "traffic": None,
                "ms_transform_chain": round(sum(ms) / len(ms), 4) if ms else None,
                "sizes_after_chain": list(self.last_sizes), "kernels": []}

    def northstar(self, peak_gbs, reps=5):
        return None

    def describe(self, scene, sizes, graph="random"):
        return (f"{self.name}; raw synthetic NAG scene {scene} (N0,N1,N2,E1,E2,clouds)={sizes} -> "
                f"{list(self.last_sizes)} nodes per level after the chain; every step re-runs the chain on a "
                "fresh copy of the raw batch")


def build(nag, dev, world=1, stages="all", mode="train", model="spt64", kernel_timers=False):
    if stages == "scatter":
        return ScatterChain(nag, dev, world)
    if mode == "infer":
        return SPTInferStep(nag, dev, world, model=model, kernel_timers=kernel_timers)
    if mode == "panoptic":
        return SPTPanopticStep(nag, dev, world, model=model, kernel_timers=kernel_timers)
    if mode == "iteration":
        return SPTIterationStep(nag, dev, world, model=model, kernel_timers=kernel_timers)
    return SPTTrainStep(nag, dev, world, model=model, kernel_timers=kernel_timers)
