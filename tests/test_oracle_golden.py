"""CPU suite: the oracle (oracle/spt_oracle.py) against fixtures produced by
the reference's own modules (tests/golden/make_golden.py).  float64 both
sides, so tolerances are tight."""
import numpy as np
import torch

from conftest import demo_nag, load_golden, t64, tl
from oracle import spt_oracle as O

TOL = dict(rtol=1e-9, atol=1e-10)


def _params(g, prefix="p__"):
    return {k[len(prefix):]: t64(v) for k, v in g.items() if k.startswith(prefix)}


def _attention_case(name):
    g = load_golden(name)
    p = {k: v.clone().requires_grad_() for k, v in _params(g).items()}
    x = t64(g["x"]).requires_grad_()
    ea = t64(g["edge_attr"]).requires_grad_()
    out = O.self_attention(x, tl(g["edge_index"]), ea, p,
                           int(g["num_heads"]), int(g["qk_dim"]))
    torch.testing.assert_close(out, t64(g["out"]), **TOL)
    (out * t64(g["gw"])).sum().backward()
    torch.testing.assert_close(x.grad, t64(g["g_x"]), **TOL)
    torch.testing.assert_close(ea.grad, t64(g["g_edge_attr"]), **TOL)
    for k, v in p.items():
        torch.testing.assert_close(v.grad, t64(g["g__" + k]), **TOL)


def test_attention_spt64_matches_reference_block():
    _attention_case("attention_spt64.npz")


def test_attention_spt128_matches_reference_block():
    _attention_case("attention_spt128.npz")


def test_unit_sphere_norm_matches_reference():
    g = load_golden("unit_sphere_norm.npz")
    pos, idx, w, ns = t64(g["pos"]), tl(g["idx"]), tl(g["w"]), int(g["num_super"])
    for key, kw in (("unw", dict(idx=idx, w=None, num_super=ns)),
                    ("w", dict(idx=idx, w=w, num_super=ns)),
                    ("none_w", dict(idx=None, w=w)),
                    ("none", dict(idx=None, w=None))):
        o, d = O.unit_sphere_norm(pos, **kw)
        torch.testing.assert_close(o, t64(g["out_" + key]), **TOL)
        torch.testing.assert_close(d, t64(g["diam_" + key]), **TOL)


def test_knn_brute_force_matches_reference():
    g = load_golden("knn_brute_force.npz")
    xyz = torch.from_numpy(g["xyz"])
    k, r = int(g["k"]), float(g["r_max"])
    nb, d = O.knn_brute_force(xyz, xyz, k + 1, r)
    assert torch.equal(nb[:, 1:], tl(g["neighbors"]))
    torch.testing.assert_close(d[:, 1:], torch.from_numpy(g["distances"]))


def test_knn_contract_agrees_with_reference_brute_force():
    """The FRNN-contract kNN (squared dists, strict radius, index ties) picks
    the same neighbours as the reference's brute force wherever no candidate
    sits exactly on the radius or ties in distance."""
    g = load_golden("knn_brute_force.npz")
    xyz = torch.from_numpy(g["xyz"])
    k, r = int(g["k"]), float(g["r_max"])
    idx, dist = O.knn_1(xyz, k, r)
    ref = tl(g["neighbors"])
    assert torch.equal(idx, ref)
    refd = torch.from_numpy(g["distances"])
    ok = ref >= 0
    torch.testing.assert_close(dist[ok].sqrt(), refd[ok], rtol=1e-5, atol=1e-6)
    assert (dist[~ok] == -1).all()


def test_geometric_features_match_reference():
    g = load_golden("geometric_features.npz")
    xyz = t64(g["xyz"])
    f = O.geometric_features(xyz, tl(g["nn"]), k_min=int(g["k_min"]))
    ref = t64(g["feats"])
    ev = t64(g["eigenval"])
    # eigenvectors are only defined away from degenerate spectra
    l = ev.clamp(min=0).sqrt()
    gap = torch.minimum(l[:, 2] - l[:, 1], l[:, 1] - l[:, 0]) > 1e-6 * l[:, 2]
    scal = [0, 1, 2, 7, 8, 9, 10]
    torch.testing.assert_close(f[:, scal], ref[:, scal], rtol=1e-7, atol=1e-9)
    vec = [3, 4, 5, 6]
    torch.testing.assert_close(f[gap][:, vec], ref[gap][:, vec], rtol=1e-6, atol=1e-8)
    assert gap.float().mean() > 0.9


def test_csr_view_matches_reference_cluster_on_demo_nag():
    """nag[i+1].sub (Cluster.pointers/points, written by the reference's own
    preprocessing) IS a CSR of nag[i].super_index."""
    lv = demo_nag()
    for i in range(3):
        si = tl(lv[i]["super_index"])
        n_sup = lv[i + 1]["pos"].shape[0]
        perm, rowptr = O.csr_view(si, n_sup)
        assert np.array_equal(rowptr.numpy(), lv[i + 1]["sub_pointers"].astype(np.int32))
        # same membership per segment (the reference's order inside a
        # segment is arbitrary; ours is ascending = stable)
        ref_pts = lv[i + 1]["sub_points"].astype(np.int64)
        ptr = rowptr.numpy().astype(np.int64)
        seg = np.repeat(np.arange(n_sup), ptr[1:] - ptr[:-1])
        ref_sorted = ref_pts[np.lexsort((ref_pts, seg))]
        assert np.array_equal(perm.numpy().astype(np.int64), ref_sorted)


def test_get_sub_size_on_demo_nag():
    lv = demo_nag()
    sis = [tl(lv[i]["super_index"]) for i in range(3)]
    sizes = O.get_sub_size(sis)
    # expected, straight from the reference-written Cluster CSR of each level
    prev = None
    for i, s in enumerate(sizes):
        ptr = lv[i + 1]["sub_pointers"].astype(np.int64)
        pts = lv[i + 1]["sub_points"].astype(np.int64)
        w = np.ones(len(pts), dtype=np.int64) if prev is None else prev[pts]
        exp = np.add.reduceat(w, ptr[:-1])
        assert np.array_equal(s.numpy(), exp)
        prev = exp
    assert int(sizes[-1].sum()) == sis[0].numel()


def test_scatter_minmax_tie_and_empty_rules():
    src = torch.tensor([[1., 5.], [3., 5.], [3., 2.], [0., 9.]])
    idx = torch.tensor([2, 2, 2, 0])
    out, arg = O.scatter_max(src, idx, dim_size=4)
    assert out.tolist() == [[0., 9.], [0., 0.], [3., 5.], [0., 0.]]
    assert arg.tolist() == [[3, 3], [4, 4], [1, 0], [4, 4]]
    out, arg = O.scatter_min(src, idx, dim_size=3)
    assert out.tolist() == [[0., 9.], [0., 0.], [1., 2.]]
    assert arg.tolist() == [[3, 3], [4, 4], [0, 2]]
    g = O.scatter_max_grad(torch.ones(4, 2), O.scatter_max(src, idx, dim_size=4)[1], 4)
    assert g.tolist() == [[0., 1.], [1., 0.], [0., 0.], [1., 1.]]


def _stage_mirror(g, kind):
    """Compose the oracle pieces exactly like the reference's Stage stack
    (src/nn/stage.py:215-286,413-444,545-571; transformer.py:195-256;
    mlp.py:85-94) and compare with the reference-run fixture."""
    p = {k: v.clone().requires_grad_() for k, v in _params(g).items()}
    norm_index = tl(g["norm_index"])
    pos, node_size = t64(g["pos"]), tl(g["node_size"])
    super_index, ei, ea = tl(g["super_index"]), tl(g["edge_index"]), t64(g["edge_attr"])
    leaf = {}
    if kind == "down":
        xc = t64(g["x_child"]).requires_grad_()
        leaf["g_x_child"] = xc
        pooled = O.pool(xc, tl(g["pool_index"]), int(g["num_super"]), "max")
        x = torch.cat((t64(g["x_parent"]), pooled), dim=1)
    else:
        xc = t64(g["x_child"]).requires_grad_()
        xp = t64(g["x_parent"]).requires_grad_()
        leaf["g_x_child"], leaf["g_x_parent"] = xc, xp
        x = torch.cat((xc, O.index_unpool(xp, tl(g["unpool_index"]))), dim=1)
    npos, diam = O.unit_sphere_norm(pos, super_index, w=node_size)
    x = torch.cat((npos, x), dim=1)

    def gn(x, pre):
        return O.graph_norm(x, norm_index, p[pre + ".weight"], p[pre + ".bias"],
                            p[pre + ".mean_scale"])

    lrelu = torch.nn.functional.leaky_relu
    i = 0
    while f"in_mlp.mlp.{i}.weight" in p:
        x = x @ p[f"in_mlp.mlp.{i}.weight"].t()
        x = lrelu(gn(x, f"in_mlp.mlp.{i + 1}"))
        i += 3
    b = 0
    while f"transformer_blocks.{b}.sa.qkv.weight" in p:
        pre = f"transformer_blocks.{b}."
        sa = {k[len(pre + "sa."):]: v for k, v in p.items() if k.startswith(pre + "sa.")}
        h = O.self_attention(gn(x, pre + "sa_norm"), ei, ea, sa, 16, 4)
        x = x + h
        if pre + "ffn.mlp.0.weight" in p:
            h = gn(x, pre + "ffn_norm")
            h = lrelu(h @ p[pre + "ffn.mlp.0.weight"].t() + p[pre + "ffn.mlp.0.bias"])
            h = h @ p[pre + "ffn.mlp.2.weight"].t() + p[pre + "ffn.mlp.2.bias"]
            x = x + h
        b += 1
    torch.testing.assert_close(x, t64(g["out"]), **TOL)
    (x * t64(g["gw"])).sum().backward()
    for k, v in leaf.items():
        torch.testing.assert_close(v.grad, t64(g[k]), **TOL)
    for k, v in p.items():
        torch.testing.assert_close(v.grad, t64(g["g__" + k]), rtol=1e-8, atol=1e-9)
    return diam


def test_down_stage_composition_matches_reference():
    g = load_golden("down_stage.npz")
    diam = _stage_mirror(g, "down")
    torch.testing.assert_close(diam, t64(g["diameter"]), **TOL)


def test_up_stage_composition_matches_reference():
    _stage_mirror(load_golden("up_stage.npz"), "up")


def test_attentive_pools_against_the_reference_modules():
    """src/nn/pool.py:84-360 (fixture: tests/golden/make_golden_pool.py, float64)."""
    G = load_golden("attentive_pool.npz")
    cases = {"a": dict(H=16, D=4, spec=None, share=False),
             "b": dict(H=4, D=8, spec="d+g", share=True),
             "c": dict(H=8, D=2, spec=0.7, share=False)}
    for tag, c in cases.items():
        p = {k[len(tag) + 5:]: t64(G[k]) for k in G if k.startswith(tag + "__p__")}
        xp, index = t64(G[f"{tag}__x_parent"]), tl(G[f"{tag}__index"])
        if "q.weight" in p:
            query = xp @ p["q.weight"].t() + p["q.bias"]
        else:
            query = p["q"].repeat(xp.shape[0], 1)
        out = O.attentive_pool(t64(G[f"{tag}__x_child"]), query, index,
                               t64(G[f"{tag}__edge_attr"]), p, c["H"], c["D"], xp.shape[0],
                               c["spec"], c["share"])
        torch.testing.assert_close(out, t64(G[f"{tag}__out"]), rtol=1e-10, atol=1e-10)


def test_optimal_neighbourhood_features_match_reference():
    """geometry.py:248-338 with k_step >= 0 (fixture: make_golden_geof_optimal.py)."""
    g = load_golden("geometric_features_optimal.npz")
    xyz, nn = t64(g["xyz"]), tl(g["nn"])
    for tag in "abc":
        k_min, k_step, k_search = (int(v) for v in g[f"{tag}_cfg"])
        f = O.geometric_features_optimal(xyz, nn, k_min, k_step, k_search)
        ref = t64(g[f"{tag}_feats"])
        scal = [0, 1, 2, 7, 8, 9, 10]
        torch.testing.assert_close(f[:, scal], ref[:, scal], rtol=1e-7, atol=1e-9)
        # normals / verticality: away from repeated eigenvalues of the CHOSEN neighbourhood
        # (linearity or planarity at 0 = two equal eigenvalues)
        gap = (ref[:, 0] > 1e-6) & (ref[:, 1] > 1e-6)
        torch.testing.assert_close(f[gap][:, 3:7], ref[gap][:, 3:7], rtol=1e-6, atol=1e-8)
        assert gap.float().mean() > 0.8
    # the search does something: sizes other than the largest get picked
    full = O.geometric_features(xyz, nn, 3)
    assert ((full[:, 7] - t64(g["a_feats"])[:, 7]).abs() > 1e-6).float().mean() > 0.2
