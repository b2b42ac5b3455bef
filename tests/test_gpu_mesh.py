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
