"""GPU parity: on-device quadtree + adjacency vs golden vectors from the reference (bit-exact indices)."""
import glob
import os

import numpy as np
import pytest
import torch

from helpers import GOLDEN, close, dev, dist_from_05, golden, mesh_from_golden_graph

pytestmark = pytest.mark.gpu


def test_kat_quadtree():
    from model.graph_functions import quadtree_decompose
    k = golden('kat.npz')
    for i in (1, 2, 3):
        lab = quadtree_decompose(k[f'kat{i}_img'], thresh=.5, max_size=4)
        assert np.array_equal(lab, k[f'kat{i}_labels']), f'KAT-{i}'
    for cond in ('max_larger_than', 'max_smaller_than', 'min_larger_than', 'min_smaller_than'):
        lab = quadtree_decompose(k['kat6_img'], thresh=.9 if 'max' in cond else .1, max_size=8, condition=cond)
        assert np.array_equal(lab, k['kat6_' + cond]), cond


@pytest.mark.parametrize('path', sorted(glob.glob(os.path.join(GOLDEN, 'graph_*.npz'))),
                         ids=lambda p: os.path.basename(p)[6:-4])
def test_graph_build(path):
    g = np.load(path, allow_pickle=False)
    mesh = mesh_from_golden_graph(g)
    assert mesh.N == len(g['npix'])
    assert np.array_equal(mesh.labels[0].cpu().numpy(), g['labels'])
    assert np.array_equal(mesh.npix.cpu().numpy(), g['npix'])
    ei = mesh.edge_index(True).cpu().numpy()
    assert np.array_equal(ei, g['edges'])
    attrs = mesh.edge_attrs(bool(g['use_attrs']))
    close(attrs, g['attrs'], atol=2e-5)


@pytest.mark.parametrize('name', ['64_1blob_clean', '96_ice_masked', '100_2blob_clean'])
def test_image_to_graph_api(name):
    from model.graph_functions import image_to_graph
    from model.utils import add_positional_encoding
    g = golden(f'graph_{name}.npz')
    x = add_positional_encoding(torch.from_numpy(g['x']).to(dev()))
    out = image_to_graph(x, thresh=float(g['thresh']), mask=g['mask'] if 'mask' in g.files else None,
                         high_interest_region=g['hir'] if 'hir' in g.files else None,
                         transform_func=dist_from_05 if bool(g['has_transform']) else None,
                         condition=str(g['condition']), use_edge_attrs=bool(g['use_attrs']))
    close(out['data'], g['data'])
    assert np.array_equal(out['edge_index'].cpu().numpy(), g['edges'])
    assert np.array_equal(out['n_pixels_per_node'].cpu().numpy(), g['npix'])


def test_batched_mesh_equals_single_clips():
    """B clips in one launch give, per clip, exactly the single-clip labels shifted by node_off."""
    from qtmpnn import synthetic
    from qtmpnn.mesh import build_mesh
    clips = [synthetic.make_clip(100 + i, n_frames=1, pixel_noise=0.0 if i % 2 else 0.05)[0, ..., 0] for i in range(5)]
    batch = torch.from_numpy(np.stack(clips)).to(dev())
    mb = build_mesh(src=batch, thresh=0.1)
    off = mb.node_off.cpu().numpy()
    assert off[0] == 0 and off[-1] == mb.N
    for i, c in enumerate(clips):
        ms = build_mesh(src=torch.from_numpy(c[None]).to(dev()), thresh=0.1)
        lab = mb.labels[i].cpu().numpy()
        assert np.array_equal(lab - off[i], ms.labels[0].cpu().numpy())
        assert off[i + 1] - off[i] == ms.N
        e1 = mb.edge_index(True).cpu().numpy()
        sel = (e1[0] >= off[i]) & (e1[0] < off[i + 1])
        assert np.array_equal(e1[:, sel] - off[i], ms.edge_index(True).cpu().numpy())


def test_mesh_from_node_values_equals_mesh_from_image():
    """Remesh input mode: nodeval + old labels must equal building from the un-flattened image."""
    from qtmpnn import ops, synthetic
    from qtmpnn.mesh import build_mesh
    img = torch.from_numpy(np.stack([synthetic.make_clip(7 + i, n_frames=1, pixel_noise=0.02)[0, ..., 0] for i in range(3)])).to(dev())
    old = build_mesh(src=img, thresh=0.1)
    val = torch.rand(old.N, 1, device=dev()) * 0.3
    m1 = build_mesh(prev=(val[:, 0], old), thresh=0.1)
    im = ops.gather_pixels(val, old).view(3, 64, 64)
    m2 = build_mesh(src=im, thresh=0.1)
    assert m1.N == m2.N and torch.equal(m1.labels, m2.labels) and torch.equal(m1.col[:m1.E], m2.col[:m2.E])


def test_full_size_properties():
    """BASELINE config-2 size (B=32, 64x64, noise 0.05): size-independent invariants of mesh and CSR."""
    from qtmpnn import synthetic
    from qtmpnn.mesh import build_mesh
    x, _ = synthetic.make_batch(2, 0, 32, 10, 1, n_digits=2, pixel_noise=0.05)
    img0 = torch.from_numpy(x[..., 0]).to(dev()).amax(dim=1)
    mesh = build_mesh(src=img0, thresh=0.1)
    lab = mesh.labels
    assert int(lab.min()) == 0 and int(lab.max()) == mesh.N - 1
    assert float(mesh.npix.sum()) == 32 * 64 * 64                     # every pixel in exactly one cell
    counts = torch.bincount(lab.reshape(-1).long(), minlength=mesh.N).float()
    assert torch.equal(counts, mesh.npix)
    # labels are clip-contiguous and DFS-ordered: the bottom-right pixel of every clip holds its first label
    off = mesh.node_off.long()
    assert torch.equal(lab[:, -1, -1].long(), off[:-1])
    # CSR is symmetric with symmetric weights, no self pairs, all weights > 0
    ei = mesh.edge_index(False)
    key = ei[0] * mesh.N + ei[1]
    assert torch.equal(torch.sort(key).values, torch.sort(ei[1] * mesh.N + ei[0]).values)
    assert bool((ei[0] != ei[1]).all()) and bool((mesh.w[:mesh.E] > 0).all())
    # L^ 1 = -D^-1/2 W D^-1/2 1 has entries in [-deg_max, 0]; and row sums of W D^-1 are 1
    rp = mesh.rowptr.long()
    src = torch.repeat_interleave(torch.arange(mesh.N, device=dev()), rp[1:] - rp[:-1])
    deg = torch.zeros(mesh.N, device=dev()).index_add_(0, src, mesh.w[:mesh.E])
    close(mesh.dis, deg.rsqrt(), rtol=1e-5)


def test_dense_mapping_compatibility():
    """the Mesh stands in for the reference's dense (N, P) `mapping` where its tooling touches it --
    `.to_dense()`, `.reshape(-1, *image_shape)` (notebooks/create_mesh.ipynb), and `flatten` / `unflatten` with a mask
    (ice_results.py:116-118): dense-matrix flatten == the kernel path."""
    from model.graph_functions import flatten, unflatten
    g = np.load(os.path.join(GOLDEN, 'graph_96_ice_masked.npz'), allow_pickle=False)
    mesh = mesh_from_golden_graph(g)
    dense = mesh.to_dense()
    lab = torch.from_numpy(g['labels'].reshape(-1)).to(dev())
    assert dense.shape == (mesh.N, lab.numel())
    assert torch.equal(dense.sum(0), (lab >= 0).float()) and torch.equal(dense.sum(1), mesh.npix)
    planes = mesh.reshape(-1, *g['labels'].shape)
    assert planes.shape == (mesh.N, *g['labels'].shape) and torch.equal(planes.reshape(mesh.N, -1), dense)
    torch.manual_seed(0)
    img = torch.rand(3, *g['labels'].shape, 2, device=dev())
    mask = g['mask'] if 'mask' in g.files else None
    flat = flatten(img, mesh, mesh.npix, mask)                                   # (3, N, 2) node means
    want = torch.einsum('np,spc->snc', dense, img.reshape(3, -1, 2)) / mesh.npix.view(1, -1, 1)
    close(flat, want, atol=1e-5)
    back = unflatten(flat, mesh, g['labels'].shape, mask)
    inside = (lab >= 0).view(*g['labels'].shape)
    close(back[:, inside], torch.einsum('np,snc->spc', dense, flat).reshape(3, *g['labels'].shape, 2)[:, inside], atol=1e-6)


@pytest.mark.parametrize('shape,masked', [((64, 64), False), ((100, 100), True), ((64, 128), False), ((128, 128), True), ((80, 150), True)])
def test_stage1_four_workgroups_per_base_cell_equals_one(shape, masked, monkeypatch):
    """qt_quadtree_stage1 with one workgroup per 32 x 32 quadrant of a 64 x 64 base cell (the default: leaf counts per quadrant in
    DFS order, k_quadtree_stage1q) against one workgroup per base cell: labels, levels, cells, node offsets, CSR bit for bit --
    image sizes that are no multiples of 64 or 32 (quadrants partly or wholly outside the image), land mask + high-interest
    region, base cells that do not split at all (one level-6 leaf), meshes decided by node values of an old mesh, static mode."""
    from qtmpnn import synthetic
    from qtmpnn.mesh import build_mesh
    n, m = shape
    rng = np.random.default_rng(n * 1000 + m)
    img = np.zeros((3, n, m), np.float32)
    img[0, n // 3:n // 3 + 9, m // 2:m // 2 + 13] = rng.random((9, 13)).astype(np.float32)          # mostly empty: whole base cells
    img[1] = (rng.random((n, m)) < 0.03).astype(np.float32)                                            # scattered pixels
    img[2] = rng.random((n, m)).astype(np.float32) * 0.2                                               # noise around the threshold
    mask = hir = None
    if masked:
        mask = np.zeros((n, m), bool)
        mask[n // 2:n // 2 + 7, 3:m // 3] = True
        hir = np.zeros((n, m), bool)
        hir[5:9, m - 12:m - 2] = True
    src = torch.from_numpy(img).to(dev())

    def build(quads, static, prev=None):
        monkeypatch.setenv('QT_NO_STAGE1_QUADS', '0' if quads else '1')
        if prev is None:
            return build_mesh(src=src, thresh=0.1, mask=mask, high_interest_region=hir, static=static)
        return build_mesh(prev=prev, thresh=0.1, mask=mask, high_interest_region=hir, static=static)
    for static in (False, True):
        a, b = build(True, static), build(False, static)
        nv = a.n_valid
        assert nv == b.n_valid and a.N == b.N
        for name in ('labels', 'level', 'node_off'):
            assert torch.equal(getattr(a, name), getattr(b, name)), name
        assert torch.equal(a.cell[:nv], b.cell[:nv]) and torch.equal(a.rowptr[:nv + 1], b.rowptr[:nv + 1])
        E = int(a.rowptr[nv])
        assert torch.equal(a.col[:E], b.col[:E]) and torch.equal(a.nrm[:E], b.nrm[:E])
        # a mesh decided by node values on the old mesh (the re-mesh of the rollout), both ways
        val = torch.rand(a.N, 4, device=dev()) * 0.2
        a2, b2 = build(True, static, (val[:, 0], a)), build(False, static, (val[:, 0], b))
        assert a2.n_valid == b2.n_valid and torch.equal(a2.labels, b2.labels) and torch.equal(a2.level, b2.level)
        assert torch.equal(a2.fwd_src[:a2.n_valid], b2.fwd_src[:b2.n_valid]) and torch.equal(a2.bwd_src[:nv], b2.bwd_src[:nv])


def test_direct_row_index_of_a_remesh_is_complete():
    """bwd_src (per OLD node the new node under its single pixel, -1 otherwise) is filled with -1 before the mesh build writes
    the entries it owns, so no entry is ever an uninitialised row index; the state transfer through the direct index equals the
    general path (labels -> rows) in both directions, on a masked mesh."""
    from qtmpnn import ops
    from qtmpnn.mesh import build_mesh
    rng = np.random.default_rng(5)
    img = (rng.random((2, 64, 64)) * 0.3).astype(np.float32)
    mask = np.zeros((64, 64), bool)
    mask[20:31, 8:40] = True
    old = build_mesh(src=torch.from_numpy(img).to(dev()), thresh=0.15, mask=mask)
    val = torch.rand(old.N, 4, device=dev()) * 0.3
    new = build_mesh(prev=(val[:, 0], old), thresh=0.15, mask=mask)
    bs = new.bwd_src[:old.N]
    assert int(bs.min()) >= -1 and int(bs.max()) < new.N and int(new.fwd_src[:new.N].min()) >= -1 and int(new.fwd_src[:new.N].max()) < old.N
    single_old = (old.npix == 1)
    assert bool((bs[~single_old] == -1).all())                  # multi-pixel old nodes take the general path
    state = torch.randn(old.N, 8, device=dev(), requires_grad=True)
    out = ops.remesh_transfer(state, old, new)
    g = torch.randn_like(out)
    (gs,) = torch.autograd.grad(out, state, g)
    keep_f, keep_b, new.fwd_src, new.bwd_src = new.fwd_src, new.bwd_src, None, None
    built, new.built_from = new.built_from, None
    try:
        out2 = ops.remesh_transfer(state, old, new)
        (gs2,) = torch.autograd.grad(out2, state, g)
    finally:
        new.fwd_src, new.bwd_src, new.built_from = keep_f, keep_b, built
    assert torch.equal(out, out2) and torch.equal(gs, gs2)


@pytest.mark.parametrize('noise,static', [(0.0, False), (0.05, False), (0.05, True)])
def test_tail_row_records(noise, static):
    """qt_edges_norm's tail_rec array (what csrc/chebclip.hip runs the rows with more than four edges from): clip c holds exactly
    one record per such row -- its first four CSR edges (columns relative to the clip, packed as the kernel keeps them; weights =
    nrm), its tail_info word and its row number -- and rows + records fit the kernel's 4096 row slots (a row with a tail is a
    cell of at least two pixels)."""
    from qtmpnn import synthetic
    from qtmpnn.mesh import build_mesh
    B = 3
    img = np.stack([synthetic.make_clip(20 + i, n_frames=1, pixel_noise=noise)[0, ..., 0] for i in range(B)])
    mesh = build_mesh(src=torch.from_numpy(img).to(dev()), thresh=0.1, static=static)
    off = mesh.node_off.cpu().numpy()
    rp, col, nrm = mesh.rowptr.cpu().numpy(), mesh.col.cpu().numpy(), mesh.nrm.cpu().numpy()
    rec = mesh.tail_rec.cpu().numpy().view(np.uint32)
    tinfo = mesh.tail_info.cpu().numpy().view(np.uint32)
    tcnt = mesh.tail_cnt.cpu().numpy()
    total = 0
    for c in range(B):
        r0, nr = int(off[c]), int(off[c + 1] - off[c])
        T = int(tcnt[32 * c + 1])
        deg = rp[r0 + 1:r0 + nr + 1] - rp[r0:r0 + nr]
        assert T == int((deg > 4).sum()) and nr + T <= 4096
        rows = set()
        for j in range(T):
            ent = rec[c, j]
            r = int(ent[7])
            assert r not in rows and 0 <= r < nr and deg[r] > 4
            rows.add(r)
            e0 = int(rp[r0 + r])
            cols = [int(col[e0 + k]) - r0 for k in range(4)]
            assert [(int(ent[0]) >> 4) & 4095, int(ent[0]) >> 20, (int(ent[1]) >> 4) & 4095, int(ent[1]) >> 20] == cols
            assert ent[2:6].view(np.float32).tolist() == [float(np.float32(nrm[e0 + k])) for k in range(4)]
            assert int(ent[6]) == int(tinfo[r0 + r]) != 0
        total += T
    assert total > 0
