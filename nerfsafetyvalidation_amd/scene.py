"""Deterministic synthetic "Stonehenge" workload.

The reference ships no dataset, no checkpoint and no occupancy grid, so every benchmark and
parity input is generated here from fixed seeds:

  * camera: nerf_synthetic field of view (camera_angle_x = 0.6911112070083618), pinhole intrinsics by
    the reference formula (nerf/provider.py:258-274), deterministic orbit poses built with the
    look-at construction of rand_poses (nerf/provider.py:57-91) on a fixed (theta, phi) grid;
  * occupancy: a procedural henge (ground slab + ring of upright stones + lintels) rasterised into
    the cascaded 128^3 density grid in Morton order, then packed by the packbits rule
    (raymarching.cu:286-288: bit i of byte n <=> cell 8n+i);
  * model: hash table U(-0.5, 0.5) under torch.manual_seed(0) rounded to fp16, FFMLP weights by the
    reference initialiser (seed 42), density_scale chosen so rays saturate in tens of samples.

Everything is plain numpy / torch on the host: no native code, usable on CPU.
"""
import hashlib
import math

import numpy as np
import torch

CAMERA_ANGLE_X = 0.6911112070083618


def intrinsics(H, W, camera_angle_x=CAMERA_ANGLE_X):
    fl = W / (2 * np.tan(camera_angle_x / 2))
    return np.array([fl, fl, W / 2, H / 2], dtype=np.float64)


def orbit_poses(n_theta=5, n_phi=40, radius=1.5, theta_range=(np.pi / 3, 2 * np.pi / 3)):
    """[n_theta*n_phi, 4, 4] float32 cam2world on a fixed grid (200 views by default)."""
    thetas = np.linspace(theta_range[0], theta_range[1], n_theta)
    phis = np.linspace(0, 2 * np.pi, n_phi, endpoint=False)
    tt, pp = np.meshgrid(thetas, phis, indexing="ij")
    tt, pp = tt.reshape(-1), pp.reshape(-1)
    centers = np.stack([radius * np.sin(tt) * np.sin(pp), radius * np.cos(tt), radius * np.sin(tt) * np.cos(pp)], -1)

    def normalize(v):
        return v / (np.linalg.norm(v, axis=-1, keepdims=True) + 1e-10)

    forward = -normalize(centers)
    up = np.tile(np.array([0.0, -1.0, 0.0]), (len(tt), 1))
    right = normalize(np.cross(forward, up))
    up = normalize(np.cross(right, forward))
    poses = np.tile(np.eye(4), (len(tt), 1, 1))
    poses[:, :3, :3] = np.stack([right, up, forward], -1)
    poses[:, :3, 3] = centers
    return poses.astype(np.float32)


def _expand_bits(v):
    v = (v * 0x00010001) & 0xFF0000FF
    v = (v * 0x00000101) & 0x0F00F00F
    v = (v * 0x00000011) & 0xC30C30C3
    v = (v * 0x00000005) & 0x49249249
    return v


def morton3d_np(x, y, z):
    x, y, z = [np.asarray(a, dtype=np.uint64) & 0x3FF for a in (x, y, z)]
    return (_expand_bits(x) | (_expand_bits(y) << 1) | (_expand_bits(z) << 2)).astype(np.int64) & 0xFFFFFFFF


def henge_occupancy(x, y, z):
    """boolean occupancy of world points (y up). Ground slab, 30 uprights on a radius-0.6 ring, 5 trilithons inside."""
    occ = (y > -0.36) & (y < -0.30) & (x * x + z * z < 0.95 ** 2)           # ground slab
    r = np.sqrt(x * x + z * z)
    ang = np.arctan2(z, x)
    # outer ring: 30 stones, each ~0.07 wide tangentially, 0.05 thick radially, height -0.30..0.05
    k = np.round(ang / (2 * np.pi / 30))
    dang = ang - k * (2 * np.pi / 30)
    ring = (np.abs(r - 0.6) < 0.03) & (np.abs(dang * 0.6) < 0.035) & (y >= -0.30) & (y < 0.05)
    lintel = (np.abs(r - 0.6) < 0.03) & (y >= 0.05) & (y < 0.09)              # continuous lintel ring
    occ |= ring | lintel
    # inner horseshoe: 5 trilithons (two uprights + lintel) on radius 0.3
    for j in range(5):
        a0 = np.pi * (0.15 + 0.175 * j * 2)
        cx, cz = 0.3 * np.cos(a0), 0.3 * np.sin(a0)
        tx, tz = -np.sin(a0), np.cos(a0)                                       # tangent
        dx, dz = x - cx, z - cz
        u = dx * tx + dz * tz                                                  # tangential coordinate
        v = dx * np.cos(a0) + dz * np.sin(a0)                                  # radial coordinate
        up = (np.abs(v) < 0.035) & (np.abs(np.abs(u) - 0.06) < 0.03) & (y >= -0.30) & (y < 0.15)
        top = (np.abs(v) < 0.035) & (np.abs(u) < 0.1) & (y >= 0.15) & (y < 0.2)
        occ |= up | top
    return occ


def density_grid(bound=2, grid_size=128, occupied_value=1.0):
    """[cascade, grid_size^3] float32 density grid in Morton order (nerf/renderer.py:453-544 layout)."""
    cascade = 1 + math.ceil(math.log2(bound))
    H = grid_size
    idx = np.arange(H)
    ii, jj, kk = np.meshgrid(idx, idx, idx, indexing="ij")
    mort = morton3d_np(ii.reshape(-1), jj.reshape(-1), kk.reshape(-1))
    grid = np.zeros((cascade, H ** 3), dtype=np.float32)
    for cas in range(cascade):
        mip_bound = min(2 ** cas, bound)
        c = (-1.0 + (2 * idx + 1) / H) * mip_bound                              # cell centres
        half = mip_bound / H
        occ = np.zeros((H, H, H), dtype=bool)
        # conservative rasterisation: a cell is occupied if its centre or any of its 8 corners is inside
        for ox in (-half, 0.0, half):
            for oy in (-half, 0.0, half):
                for oz in (-half, 0.0, half):
                    occ |= henge_occupancy((c + ox)[:, None, None], (c + oy)[None, :, None], (c + oz)[None, None, :])
        grid[cas, mort] = occ.reshape(-1).astype(np.float32) * occupied_value
    return grid


def packbits_np(grid, thresh):
    """bit i of byte n <=> cell 8n+i > thresh (raymarching.cu:269-291)"""
    flat = (grid.reshape(-1, 8) > thresh).astype(np.uint8)
    weights = (1 << np.arange(8)).astype(np.uint8)
    return (flat * weights).sum(-1).astype(np.uint8)


def bitfield_sha256(bitfield):
    return hashlib.sha256(np.ascontiguousarray(bitfield).tobytes()).hexdigest()


class StonehengeScene:
    """Bundles camera set + occupancy + model hyper-parameters of BASELINE.json configs[1]."""

    def __init__(self, H=800, W=800, bound=2, radius=1.5, density_scale=48.0, n_theta=5, n_phi=40):
        self.H, self.W, self.bound, self.radius = H, W, bound, radius
        self.density_scale = density_scale
        self.min_near = 0.2
        self.intrinsics = intrinsics(H, W)
        self.poses = orbit_poses(n_theta, n_phi, radius)
        self.cascade = 1 + math.ceil(math.log2(bound))
        self.grid_size = 128
        self._grid = None

    @property
    def grid(self):
        if self._grid is None:
            self._grid = density_grid(self.bound, self.grid_size)
        return self._grid

    def bitfield(self):
        return packbits_np(self.grid, 0.01)

    def build_model(self, device, backbone="ff", cuda_ray=True, table_seed=0, fp16_table=True):
        """NeRFNetwork (FFMLP backbone by default) with the synthetic table, weights and occupancy, in eval mode.
        fp16_table=False keeps the table's full fp32 draws (NOT representable in fp16: what an fp32-trained checkpoint holds)."""
        if backbone == "ff":
            from .nerf.network_ff import NeRFNetwork
        else:
            from .nerf.network import NeRFNetwork
            torch.manual_seed(0)
        model = NeRFNetwork(encoding="hashgrid", bound=self.bound, cuda_ray=cuda_ray, density_scale=self.density_scale,
                            min_near=self.min_near, density_thresh=0.01, bg_radius=-1)
        g = torch.Generator().manual_seed(table_seed)
        emb = torch.rand(model.encoder.embeddings.shape, generator=g) - 0.5
        model.encoder.embeddings.data.copy_(emb.half().float() if fp16_table else emb)
        if cuda_ray:
            model.density_grid.copy_(torch.from_numpy(self.grid))
            model.density_bitfield.copy_(torch.from_numpy(self.bitfield()))
        return model.to(device).eval()
