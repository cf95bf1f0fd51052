"""Host logic of the generic-architecture path (no GPU): the program builder behind nsos_mlp_generic_* is pure host arithmetic, so the
layout of the saved-activation rows (nsos_mlp_generic_save_layout), the packed sizes and the backward program's size can be checked on
a CPU-only box.  The description is filled from CPU tensors (their pointers are never dereferenced by these entry points)."""
import ctypes as C

import pytest
import torch

import nerf_sos_amd
from nerf_sos_amd import _lib, ops
from helpers import GENERIC_CASES

EXTRA = {
    "shipped": dict(use_semantics=True, sem_with_coord=True),
    "d16w64_skips": dict(netdepth=16, netwidth=64, netdepth_fine=16, netwidth_fine=64),
    "w768": dict(netwidth=768, netwidth_fine=768, use_semantics=True, sem_with_coord=True),       # 16-point tiles, two-Linear head
    "w512": dict(netwidth=512, netwidth_fine=512),                                                  # 32-point tiles still fit
}


def _plan(kwargs, monkeypatch):
    monkeypatch.setattr(ops, "_dev", lambda t, name: t)
    net = nerf_sos_amd.NeRFNet(**kwargs)
    m = net.nerf
    return m, ops.GenericPlan(m.mlp, m.multires, m.multires_views)


@pytest.mark.parametrize("name", list(GENERIC_CASES) + list(EXTRA))
def test_saved_row_layout_is_a_partition(name, monkeypatch):
    """Every Linear of the module tree appears exactly once, in forward order; column blocks are 32-aligned, disjoint and tile
    [x block | v block | Linear blocks] = ld exactly; a Linear's segments cover its weight's columns; every segment's source block is
    the encodings' or an EARLIER Linear's block wide enough for its rows."""
    kwargs = GENERIC_CASES[name][0] if name in GENERIC_CASES else EXTRA[name]
    m, plan = _plan(kwargs, monkeypatch)
    ld, layout = plan.layout()
    linears = [n for n, mod in m.mlp.named_modules() if isinstance(mod, torch.nn.Linear)]
    assert sorted(n for n, *_ in layout) == sorted(linears)
    params = dict(m.mlp.named_parameters())
    pad = lambda n: (n + 31) // 32 * 32  # noqa: E731
    x_dim = m.mlp.input_ch
    v_dim = m.mlp.input_ch_views if m.mlp.use_viewdirs else 0
    blocks = {0: pad(x_dim)}
    if v_dim:
        blocks[pad(x_dim)] = pad(v_dim)
    end = pad(x_dim) + pad(v_dim)
    for name_, col, out_dim, segs in layout:
        w = params[name_ + ".weight"]
        assert col == end and col % 32 == 0 and out_dim == w.shape[0], (name_, col, end)
        covered = 0
        for src_col, rows, wcol in segs:
            assert wcol == covered and src_col in blocks and blocks[src_col] >= pad(rows) and src_col < col, (name_, src_col, rows)
            covered += rows
        assert covered == w.shape[1], (name_, covered, w.shape)
        blocks[col] = pad(out_dim)
        end = col + pad(out_dim)
    # ... followed by the ReLU patterns as bit words (round 5): one 32-bit word per output tile of every ReLU Linear, the row padded to 16 bytes
    sem = [n for n in linears if n.startswith("semantic_linear.")]
    relu = [n for n in linears if n.startswith("pts_linears.") or n == "views_linears.0" or n == "geo_map_sem.0" or (n in sem and n != sem[-1])]
    if not (m.mlp.use_viewdirs and sem):
        relu = [n for n in relu if not n.startswith(("semantic_linear.", "geo_map_sem."))]    # the head never runs without view directions
    words = sum(pad(params[n + ".weight"].shape[0]) // 32 for n in relu)
    assert ld == (end + words + 3) // 4 * 4, (ld, end, words)
    lib = _lib.lib()
    fwd = lib.nsos_mlp_generic_packed_bytes(C.byref(plan.desc))
    bwd = lib.nsos_mlp_generic_bwd_packed_bytes(C.byref(plan.desc), 0)
    bwd_in = lib.nsos_mlp_generic_bwd_packed_bytes(C.byref(plan.desc), 1)
    assert fwd > 0 and 0 < bwd < bwd_in                 # the program that reaches the encodings carries their transposed streams on top
    assert plan.out_channels == (4 + (m.mlp.semantic_linear[-1].weight.shape[0] if (m.mlp.use_semantics and m.mlp.use_viewdirs) else 0))


def test_layout_capacity_and_null_checks():
    lib = _lib.lib()
    G = _lib.GenericMlp()
    table = (C.c_int32 * 4)()
    assert lib.nsos_mlp_generic_save_layout(C.byref(G), table, 4) < 0          # an empty description is refused, not dereferenced
    assert lib.nsos_mlp_generic_bwd_packed_bytes(C.byref(G), 0) == 0
    assert lib.nsos_mlp_generic_save_layout(None, table, 4) < 0


def test_widths_beyond_the_lds_budget_are_refused(monkeypatch):
    """32-point tiles to W = 576 (384 with a deep semantic head), 16-point tiles to 800 (include/nerf_sos_hip.h); beyond that the
    constructor succeeds (parameters only) and the first pack raises NotImplementedError -- nothing renders wrongly."""
    for kwargs in (dict(netwidth=1024, netwidth_fine=1024), dict(netwidth=832, netwidth_fine=832, use_semantics=True, sem_layer=4)):
        with pytest.raises(NotImplementedError):
            _plan(kwargs, monkeypatch)
    _plan(dict(netwidth=800, netwidth_fine=800, use_semantics=True, sem_layer=4), monkeypatch)
    _plan(dict(netwidth=800, netwidth_fine=800, use_semantics=True, sem_with_coord=True), monkeypatch)


def test_semantic_channel_limit_and_repeated_skips(monkeypatch):
    """sem_dim 9 .. 28 used to pass the program builder's row check and overflow the OUT buffer in LDS; the limit is 8
    (the output tile's one k-group of logits = nsos_composite_backward's channel limit) and both the Python plan and the C entry
    refuse more.  A repeated skip index means what `i in self.skips` means in the reference (models/nerf_mlp.py:73): once."""
    _plan(dict(use_semantics=True, sem_dim=8), monkeypatch)
    for sem_dim in (9, 13, 16, 28):
        with pytest.raises(NotImplementedError):
            _plan(dict(use_semantics=True, sem_dim=sem_dim), monkeypatch)
    m, plan = _plan(dict(use_semantics=True, sem_dim=8), monkeypatch)
    plan.desc.sem_dim = 16                                   # straight at the C ABI: the builder itself refuses
    assert _lib.lib().nsos_mlp_generic_packed_bytes(C.byref(plan.desc)) == 0
    monkeypatch.setattr(ops, "_dev", lambda t, name: t)
    m1 = nerf_sos_amd.NeRFMLP(net_depth=6, skips=[2, 2, 4])
    p1 = ops.GenericPlan(m1.mlp, m1.multires, m1.multires_views)
    assert p1.desc.skip_mask == (1 << 2) | (1 << 4)
