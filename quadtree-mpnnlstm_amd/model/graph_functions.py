"""Mesh construction and mesh<->image transfers: the reference's model/graph_functions.py
surface on top of the on-device mesh builder (qtmpnn.mesh) -- same function names, argument
meaning and error behaviour, with the dense (N, P) `mapping` replaced by a `Mesh` label map.

Citations are into the reference tree.
"""
import warnings

import numpy as np
import torch

from qtmpnn import ops
from qtmpnn._lib import on_device
from qtmpnn.mesh import CONDITIONS as _CONDITIONS, Mesh, build_mesh, build_pixel_mesh, host_mask

CONDITIONS = list(_CONDITIONS)


class _Bag:
    """Stand-in for the torch_geometric Data object the reference stores in Graph.pyg."""

    def __init__(self, **kw):
        self.__dict__.update(kw)

    def to(self, *_a, **_k):
        return self


class Graph:
    """Mesh state of one rollout (model/graph_functions.py:23-33)."""

    def __init__(self, edge_index, edge_attr, **kwargs):
        self.pyg = _Bag(edge_index=edge_index, edge_attr=edge_attr, **kwargs)
        self.mapping = None
        self.n_pixels_per_node = None
        self.hidden = None
        self.cell = None


def _device_of(*xs):
    for x in xs:
        if torch.is_tensor(x) and x.is_cuda:
            return x.device
    if not torch.cuda.is_available():
        raise RuntimeError('the qtmpnn graph builder runs on the GPU only (no CPU fallback)')
    return torch.device('cuda', torch.cuda.current_device())


def _criterion(img0, n, m, max_size, transform_func):
    """Edge-pad to the base grid and apply transform_func (graph_functions.py:190-194); None if no transform."""
    if transform_func is None:
        return img0
    n_pad, m_pad = -(n // -max_size) * max_size, -(m // -max_size) * max_size
    padded = torch.nn.functional.pad(img0.unsqueeze(1), (0, m_pad - m, 0, n_pad - n), mode='replicate').squeeze(1)
    return transform_func(padded)


def quadtree_decompose(img, padding=0, thresh=0.05, max_size=8, mask=None, high_interest_region=None,
                       transform_func=None, condition='max_larger_than'):
    """Label every pixel with its quadtree leaf (graph_functions.py:145-259); returns an int64 array,
    -1 = masked.  `img` is a 2-D array or tensor; the work happens on the GPU."""
    assert max_size & (max_size - 1) == 0, f'max_size / max_grid_size = {max_size}: must be a power of two'
    assert condition in CONDITIONS
    assert padding == 0, 'padding is unused by the reference call sites'
    dev = _device_of(img)
    t = torch.as_tensor(np.asarray(img) if not torch.is_tensor(img) else img).to(dev).float()
    n, m = t.shape
    crit = _criterion(t.unsqueeze(0), n, m, max_size, transform_func)
    mesh = build_mesh(src=crit, n=n, m=m, thresh=thresh, condition=condition, mask=mask,
                      high_interest_region=high_interest_region, max_size=max_size)
    return mesh.labels[0].long().cpu().numpy()


def get_mapping(labels):
    """Sparse (N, P) one-hot mapping, node ids and pixel counts of a label image (graph_functions.py:555-587)."""
    flat = torch.as_tensor(np.asarray(labels)).reshape(-1).long()
    keep = flat >= 0
    rows, cols = flat[keep], torch.nonzero(keep).flatten()
    n_nodes = int(rows.max()) + 1 if rows.numel() else 0
    mapping = torch.sparse_coo_tensor(torch.stack([rows, cols]), torch.ones(rows.numel()), size=(n_nodes, flat.numel()))
    return mapping, np.arange(n_nodes), torch.bincount(rows, minlength=n_nodes).float()


def get_adj(labels, xx, yy, edges_at_corners=False, use_edge_attrs=True):
    """Directed 4-neighbour edges between cells incl. self pairs of multi-pixel cells, in canonical
    (source, target) order (graph_functions.py:261-356; the reference order is scan x set order)."""
    assert not edges_at_corners, 'edges_at_corners is unused by the reference'
    lab = torch.as_tensor(np.asarray(labels)).long()
    pairs = []
    for a, b in ((lab[:-1], lab[1:]), (lab[:, :-1], lab[:, 1:])):
        a, b = a.reshape(-1), b.reshape(-1)
        ok = (a >= 0) & (b >= 0)
        pairs += [torch.stack([a[ok], b[ok]]), torch.stack([b[ok], a[ok]])]
    e = torch.cat(pairs, dim=1)
    n_nodes = int(lab.max()) + 1
    key = torch.unique(e[0] * n_nodes + e[1])
    ei = torch.stack([key // n_nodes, key % n_nodes])
    xx, yy = torch.as_tensor(xx).cpu(), torch.as_tensor(yy).cpu()
    d = dist(ei[0], ei[1], xx, yy)
    attrs = torch.stack((dist_angle(ei[0], ei[1], xx, yy), d)).T if use_edge_attrs else d
    return ei, attrs


def dist(node0, node1, xx, yy):
    """Centroid distance (graph_functions.py:358-363)."""
    return torch.sqrt((yy[node0] - yy[node1]) ** 2 + (xx[node0] - xx[node1]) ** 2)


def dist_angle(node0, node1, xx, yy):
    """Bearing in [0, 1) (graph_functions.py:365-370)."""
    return torch.atan2(xx[node0] - xx[node1], yy[node0] - yy[node1]) % (2 * np.pi) / (2 * np.pi)


def flatten_pixelwise(img, mask):
    if mask is not None:
        return img[:, ~torch.as_tensor(host_mask(mask), dtype=torch.bool, device=img.device), :]
    return img.reshape(img.shape[0], -1, img.shape[-1])


@on_device(lambda img, *a, **k: img)
def flatten(img, mapping, n_pixels_per_node, mask=None):
    """Image (n_samples, w, h, c) -> node means (n_samples, N, c) (graph_functions.py:391-419).

    `mapping` is the Mesh returned by image_to_graph; a batch (B, n_samples, w, h, c) is accepted for
    B-clip meshes.  A dense (N, P) tensor is still honoured for old call sites (plain matmul)."""
    if mapping is None:
        assert len(img.shape) == 4
        return flatten_pixelwise(img, mask)
    if not isinstance(mapping, Mesh):
        assert len(img.shape) == 4, f'array should be 4-dimensional (n_samples, w, h, c); got {img.shape}'
        ns, w, h, c = img.shape
        data = torch.moveaxis(img, -1, 0).reshape(c, ns, w * h) @ mapping.T / n_pixels_per_node
        return torch.moveaxis(data, 0, -1)
    mesh = mapping
    if img.dim() == 4:
        assert mesh.B == 1, f'array should be 5-dimensional (B, n_samples, w, h, c) for a {mesh.B}-clip mesh'
        img = img.unsqueeze(0)
    assert img.dim() == 5, f'array should be 4-dimensional (n_samples, w, h, c); got {img.shape}'
    B, ns, w, h, c = img.shape
    assert (B, w, h) == (mesh.B, mesh.n, mesh.m), 'image does not match the mesh'
    return ops.pool_image(img.reshape(B, ns, w * h, c), mesh, True)


def unflatten_pixelwise(data, mask, image_shape):
    _, c = data.shape
    if mask is None:
        return data.reshape(*image_shape, c)
    img = torch.full((*image_shape, c), float('nan'), device=data.device)
    img[~torch.as_tensor(host_mask(mask), dtype=torch.bool, device=data.device), :] = data
    return img


@on_device(lambda data, *a, **k: data)
def unflatten(data, mapping, image_shape, mask=None):
    """Node values (..., N, c) -> image (..., w, h, c) (graph_functions.py:451-458); masked pixels get 0.
    For a B-clip Mesh the result has a leading clip axis (B, ..., w, h, c)."""
    if mapping is None:
        return unflatten_pixelwise(data, mask, image_shape)
    if not isinstance(mapping, Mesh):
        d = torch.moveaxis(data, -1, 0)
        img = (d @ mapping).reshape(*d.shape[:-1], *image_shape)
        return torch.moveaxis(img, 0, -1)
    mesh = mapping
    lead = data.shape[:-2]
    N, c = data.shape[-2:]
    assert N == mesh.N, f'{N} node rows for a mesh of {mesh.N} nodes'
    flat = data.reshape(-1, N, c)
    if flat.shape[0] == 1:
        packed = flat[0]
    else:
        packed = flat.permute(1, 0, 2).reshape(N, -1)
    img = ops.gather_pixels(packed, mesh)                                     # (B, P, L*c)
    if mesh.pixelwise and mesh.mask is not None:                              # unflatten_pixelwise: NaN under the mask (:460-468)
        img = img.masked_fill((mesh.labels < 0).view(mesh.B, mesh.P, 1), float('nan'))
    img = img.reshape(mesh.B, mesh.n, mesh.m, flat.shape[0], c).permute(0, 3, 1, 2, 4)
    img = img.reshape(mesh.B, *lead, mesh.n, mesh.m, c)
    return img[0] if mesh.B == 1 else img


@on_device(lambda img, *a, **k: img)
def image_to_graph(img, thresh=0.05, max_grid_size=64, mask=None, high_interest_region=None, transform_func=None,
                   condition='max_larger_than', use_edge_attrs=True, resolution=0.25):
    """Quadtree mesh of an image stack (graph_functions.py:590-681).

    img: (n_samples, w, h, c) for one clip or (B, n_samples, w, h, c) for a batch; channel 0 drives the
    decomposition (max over samples, :632), the last two channels are the positional encoding.
    Returns the reference's dict; `mapping` is a Mesh (label map) instead of a dense matrix.
    """
    assert len(img.shape) in (4, 5), f'array should be 4-dimensional (n_samples, w, h, c); got {img.shape}'
    if torch.any(torch.isnan(img)):
        raise ValueError(f'Found NaNs in image data {torch.sum(torch.isnan(img))} / {np.prod(img.shape)}')
    single = img.dim() == 4
    x = img.unsqueeze(0) if single else img
    x = x.to(_device_of(x)).float()
    B, ns, n, m, c = x.shape
    if thresh == -np.inf:
        # image_to_graph_pixelwise (:506-539): one node per unmasked pixel, size feature = resolution^2, no self
        # pairs; `mapping` is the pixelwise Mesh where the reference returns None
        mesh = build_pixel_mesh(B, n, m, mask, x.device, resolution)
        data = ops.pool_image(x.reshape(B, ns, n * m, c), mesh, True)
        data = torch.cat([data, mesh.posfeat[:, 2:3].unsqueeze(0).expand(ns, -1, 1)], dim=-1)
        return dict(edge_index=mesh.edge_index(False), edge_attrs=mesh.edge_attrs(True, resolution) if use_edge_attrs else None,
                    data=data, graph_nodes=torch.arange(mesh.N), mapping=mesh, n_pixels_per_node=mesh.npix)
    img0 = x[..., 0].amax(dim=1).detach()
    mesh = build_mesh(src=_criterion(img0, n, m, max_grid_size, transform_func), n=n, m=m, thresh=thresh,
                      condition=condition, mask=mask, high_interest_region=high_interest_region,
                      max_size=max_grid_size, resolution=resolution)
    if thresh in (np.inf, -np.inf) or np.isinf(thresh):
        # data independent (create_static_heterogeneous_graph): can be rebuilt for any batch size
        mesh.recipe = lambda b: build_mesh(src=torch.zeros(b, n, m, device=x.device), thresh=thresh, condition=condition,
                                           mask=mask, high_interest_region=high_interest_region, max_size=max_grid_size,
                                           resolution=resolution)
    data = ops.pool_image(x.reshape(B, ns, n * m, c), mesh, True)
    if torch.any(torch.isnan(data)):
        raise ValueError(f'Found NaNs in graph data {torch.sum(torch.isnan(data))} / {np.prod(data.shape)}')
    data = torch.cat([data, mesh.posfeat[:, 2:3].unsqueeze(0).expand(ns, -1, 1)], dim=-1)
    return dict(edge_index=mesh.edge_index(True), edge_attrs=mesh.edge_attrs(use_edge_attrs, resolution), data=data,
                graph_nodes=np.arange(mesh.N), mapping=mesh, n_pixels_per_node=mesh.npix)


def create_static_heterogeneous_graph(image_shape, max_grid_size, mask, high_interest_region=None, use_edge_attrs=True,
                                      resolution=0.25, device=None):
    """Static mesh that is fine near the mask / high-interest region (graph_functions.py:683-699)."""
    from model.utils import add_positional_encoding
    arr = add_positional_encoding(torch.zeros(size=(1, *image_shape, 1), device=device or _device_of()))
    g = image_to_graph(arr, thresh=np.inf, max_grid_size=max_grid_size, mask=mask,
                       high_interest_region=high_interest_region, use_edge_attrs=use_edge_attrs, resolution=resolution)
    del g['data']
    return g


@on_device(lambda *a, device=None, **k: device)
def create_static_homogeneous_graph(image_shape, max_grid_size, mask, use_edge_attrs=True, resolution=0.25, device=None):
    """Uniform preset mesh with fully masked cells removed (graph_functions.py:707-737)."""
    from qtmpnn.mesh import build_homogeneous_mesh
    n, m = image_shape
    mesh = build_homogeneous_mesh(n, m, max_grid_size, mask, 1, device or _device_of(), resolution)
    return dict(edge_index=mesh.edge_index(True), edge_attrs=mesh.edge_attrs(use_edge_attrs, resolution),
                graph_nodes=np.arange(mesh.N), mapping=mesh, n_pixels_per_node=mesh.npix)


def plot_contours(ax, labels):
    """Draw cell borders of a label image (graph_functions.py:99-113)."""
    lab = np.asarray(labels)
    for i in range(lab.shape[0]):
        for j in range(lab.shape[1]):
            if j + 1 < lab.shape[1] and lab[i, j] != lab[i, j + 1]:
                ax.plot([j + 0.5, j + 0.5], [i - 0.5, i + 0.5], c='k', lw=0.5)
            if i + 1 < lab.shape[0] and lab[i, j] != lab[i + 1, j]:
                ax.plot([j - 0.5, j + 0.5], [i + 0.5, i + 0.5], c='k', lw=0.5)
