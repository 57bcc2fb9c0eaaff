"""Deterministic synthetic multi-view scenes (stand-ins for ETH3D / T&T, which are not in the image).

SURVEY.md section 8(d): analytic scene rendered by exact ray/plane intersection, pinhole cameras
`K = [f 0 cx; 0 f cy; 0 0 1]`, `f = 0.9 W`, procedural texture (sum of four sinusoid products) with
optional textureless rectangles + per-view noise, quantised to 8 bit like a decoded JPEG
(APD.cpp:410-413 reads uint8 and converts to float).  Works on CPU or GPU torch tensors.

The default scene (two slanted Lambertian planes every source sees completely) is the best case of the path's shortcuts.  HARD is
the other end, still analytic: slanted slabs floating in front of the planes (depth steps, occlusions: a source sees the
background where the reference sees a slab and the other way round), a per-view gain / offset, a wider camera ring whose
sources aim off the target so that parts of the reference frame project outside a source (APD.cu:546-548 returns 2.0 there).
"""
import math

import numpy as np
import torch


def _lookat_rotation(center, target):
    """World->camera rotation whose +Z axis points from `center` to `target` (y roughly down)."""
    z = np.asarray(target, np.float64) - np.asarray(center, np.float64)
    z /= np.linalg.norm(z)
    up = np.array([0.0, 1.0, 0.0])
    x = np.cross(up, z)
    x /= np.linalg.norm(x)
    y = np.cross(z, x)
    return np.stack([x, y, z], 0)


class Scene:
    """Cameras + images (+ reference ground-truth depth) of one reference view and its sources."""

    def __init__(self, width, height, num_src, images, K, R, t, depth_min, depth_max, gt_depth, view_depths=None):
        self.width, self.height, self.num_src = width, height, num_src
        self.images = images  # list of float32 [H, W] torch tensors with integer grey values
        self.K, self.R, self.t = K, R, t  # lists of float32 numpy (9,), (9,), (3,)
        self.depth_min, self.depth_max = depth_min, depth_max
        self.gt_depth = gt_depth  # float32 [H, W] torch tensor (z-depth in the reference camera)
        self.view_depths = view_depths  # with keep_view_depths: the same for EVERY view (what a converged geometric pass reads of its sources)

    def images_numpy(self):
        return [im.detach().cpu().numpy() for im in self.images]


HARD = dict(clutter=14, gain=0.10, baseline=0.10, aim_jitter=0.30)


def make_scene(width, height, num_src, seed=0, textureless=0.0, rotate=True, baseline=0.06,
               ref_view=0, device="cpu", noise=1.5, keep_view_depths=False, clutter=0, gain=0.0, aim_jitter=0.0):
    """Render view `ref_view` of a ring of cameras as the reference and `num_src` neighbours.

    textureless: fraction (0..~0.4) of the surface covered by low-texture rectangles.
    clutter: number of slanted rectangular slabs at depths 1.25..1.85 in front of the planes (depth steps + occlusions).
    gain: per-view photometric change, grey = 128 + (1 + g) * texture + 64 * o with g, o uniform in [-gain, gain] (the reference view keeps g = o = 0).
    aim_jitter: every source looks at the target displaced by up to this many scene units in x and y (sources lose overlap).
    `make_scene(..., **HARD)` is the hard preset of the bench's *_hard workloads and of the parity tests that use it.
    """
    rng = np.random.RandomState(seed)
    f = 0.9 * width
    cx, cy = 0.5 * width, 0.5 * height
    K = np.array([f, 0, cx, 0, f, cy, 0, 0, 1], np.float64)
    target = np.array([0.0, 0.0, 2.2])
    # two slanted planes n.P + d = 0 ; visible surface = nearest intersection along the ray
    planes = [(np.array([-0.15, -0.10, 1.0]), -2.0), (np.array([0.25, 0.05, 1.0]), -2.6)]
    # camera ring: view j sits at angle j on a circle of radius baseline*(1+j mod 3), ref_view shifts it
    centers, rots = [], []
    for k in range(num_src + 1):
        j = ref_view + k
        if k == 0 and ref_view == 0:
            c = np.zeros(3)
        else:
            ang = 2.399963 * j  # golden angle: well spread directions
            rad = baseline * (1 + (j % 3)) * (0.5 if k == 0 else 1.0)
            c = np.array([rad * math.cos(ang), rad * math.sin(ang), 0.02 * math.sin(1.3 * j)])
        centers.append(c)
        aim = target
        if aim_jitter > 0 and k > 0:
            jr = np.random.RandomState((7919 * seed + 31 * j + 3) % (2 ** 32))
            aim = target + np.array([jr.uniform(-aim_jitter, aim_jitter), jr.uniform(-aim_jitter, aim_jitter), 0.0])
        rots.append(_lookat_rotation(c, aim) if (rotate and not (k == 0 and ref_view == 0)) else np.eye(3))
    # texture parameters (wavelengths in reference pixels at depth 2)
    pix = 2.0 / f
    lam = np.array([5.0, 9.0, 17.0, 31.0]) * pix
    phi = rng.uniform(0, math.pi, 4)
    psi = rng.uniform(0, 2 * math.pi, (4, 2))
    amp = np.array([38.0, 30.0, 24.0, 18.0])
    rects = []
    if textureless > 0:
        nrect = 3
        side = math.sqrt(textureless / nrect) * 2.0 * (width / f)
        for _ in range(nrect):
            x0 = rng.uniform(-0.8, 0.8 - side) * (width / f)
            y0 = rng.uniform(-0.8, 0.8 - side) * (height / f)
            rects.append((x0, y0, x0 + side, y0 + side * height / width))
    # slabs: (normal, d, x0, x1, y0, y1, texture shift x, y); a bounded piece of the plane n.P + d = 0
    slabs = []
    half_w, half_h = 0.5 * width / f, 0.5 * height / f   # half extent of the frame per unit depth
    for _ in range(int(clutter)):
        z = rng.uniform(1.25, 1.85)
        sx, sy = rng.uniform(0.05, 0.22) * 2 * half_w * z, rng.uniform(0.05, 0.22) * 2 * half_h * z
        mx, my = rng.uniform(-0.95, 0.95) * half_w * z, rng.uniform(-0.95, 0.95) * half_h * z
        n = np.array([rng.uniform(-0.3, 0.3), rng.uniform(-0.3, 0.3), 1.0])
        d = -float(n @ np.array([mx, my, z]))
        slabs.append((n, d, mx - 0.5 * sx, mx + 0.5 * sx, my - 0.5 * sy, my + 0.5 * sy, rng.uniform(-3, 3), rng.uniform(-3, 3)))
    gains = [(0.0, 0.0)]
    for k in range(1, num_src + 1):
        gr = np.random.RandomState((104729 * seed + 13 * (ref_view + k) + 1) % (2 ** 32))
        gains.append((gr.uniform(-gain, gain), gr.uniform(-gain, gain)) if gain > 0 else (0.0, 0.0))

    dev = torch.device(device)
    dt = torch.float64
    ys, xs = torch.meshgrid(torch.arange(height, device=dev, dtype=dt), torch.arange(width, device=dev, dtype=dt), indexing="ij")
    images, Ks, Rs, ts = [], [], [], []
    gt_depth = None
    view_depths = [] if keep_view_depths else None
    view_normals, view_textured = ([] if keep_view_depths else None), ([] if keep_view_depths else None)
    for k in range(num_src + 1):
        Rm, c = rots[k], centers[k]
        Rt = torch.tensor(Rm.T, device=dev, dtype=dt)
        cc = torch.tensor(c, device=dev, dtype=dt)
        dcx, dcy = (xs - cx) / f, (ys - cy) / f
        # ray direction in world = R^T [dcx, dcy, 1]
        dw = [Rt[i, 0] * dcx + Rt[i, 1] * dcy + Rt[i, 2] for i in range(3)]
        best_s = None
        surf = None   # keep_view_depths: index of the surface a ray hits (planes first, then slabs)
        for pi_, (n, d) in enumerate(planes):
            num = -(float(n @ c) + d)
            den = n[0] * dw[0] + n[1] * dw[1] + n[2] * dw[2]
            s = num / den
            s = torch.where(s > 1e-6, s, torch.full_like(s, 1e9))
            if keep_view_depths:
                surf = torch.zeros_like(s, dtype=torch.int32) if best_s is None else torch.where(s < best_s, torch.full_like(surf, pi_), surf)
            best_s = s if best_s is None else torch.minimum(best_s, s)
        shift_x = shift_y = None
        for si_, (n, d, x0, x1, y0, y1, ox, oy) in enumerate(slabs):
            num = -(float(n @ c) + d)
            den = n[0] * dw[0] + n[1] * dw[1] + n[2] * dw[2]
            s = num / den
            hx, hy = cc[0] + s * dw[0], cc[1] + s * dw[1]
            hit = (s > 1e-6) & (hx >= x0) & (hx <= x1) & (hy >= y0) & (hy <= y1) & (s < best_s)
            best_s = torch.where(hit, s, best_s)
            if keep_view_depths:
                surf = torch.where(hit, torch.full_like(surf, len(planes) + si_), surf)
            if shift_x is None:
                shift_x, shift_y = torch.zeros_like(best_s), torch.zeros_like(best_s)
            shift_x = torch.where(hit, torch.full_like(shift_x, ox), shift_x)
            shift_y = torch.where(hit, torch.full_like(shift_y, oy), shift_y)
        P = [cc[i] + best_s * dw[i] for i in range(3)]
        if k == 0:
            gt_depth = best_s.to(torch.float32)  # camera-frame z: ray has z=1 in camera coordinates
        if keep_view_depths:
            view_depths.append(best_s.to(torch.float32))
            # unit normal of the surface hit, world frame, facing the camera (the convention of normals.dmb: APD.cu:1600)
            all_n = [n_ for n_, _ in planes] + [q_[0] for q_ in slabs]
            nmap = torch.zeros((height, width, 3), device=dev, dtype=torch.float32)
            for idx_, n_ in enumerate(all_n):
                u_ = np.asarray(n_, np.float64) / np.linalg.norm(n_)
                facing = (u_[0] * dw[0] + u_[1] * dw[1] + u_[2] * dw[2]) > 0     # pointing away from the camera along this ray: flip
                sel_ = surf == idx_
                for a_ in range(3):
                    toward = torch.where(facing, torch.full_like(best_s, -u_[a_]), torch.full_like(best_s, u_[a_])).to(torch.float32)
                    nmap[..., a_] = torch.where(sel_, toward, nmap[..., a_])
            view_normals.append(nmap)
        X, Y = P[0], P[1]
        if shift_x is not None:   # a slab carries the same procedural texture, displaced: no continuation across its edge
            X, Y = X + shift_x, Y + shift_y
        tex = torch.zeros_like(X)
        for q in range(4):
            u = X * math.cos(phi[q]) + Y * math.sin(phi[q])
            v = -X * math.sin(phi[q]) + Y * math.cos(phi[q])
            w = 2 * math.pi / lam[q]
            tex = tex + amp[q] * torch.sin(w * u + psi[q, 0]) * torch.sin(w * v + psi[q, 1])
        scale = torch.ones_like(X)
        for (x0, y0, x1, y1) in rects:
            inside = (X >= x0) & (X <= x1) & (Y >= y0) & (Y <= y1)
            scale = torch.where(inside, torch.full_like(scale, 0.03), scale)
        if keep_view_depths:
            view_textured.append(scale > 0.5)
        img = 128.0 + tex * scale
        if gains[k] != (0.0, 0.0):
            img = 128.0 + (1.0 + gains[k][0]) * tex * scale + 64.0 * gains[k][1]
        if noise > 0:
            g = torch.Generator(device="cpu")
            g.manual_seed(1000 * seed + 17 * (ref_view + k) + 5)
            nz = (torch.rand(height, width, generator=g, dtype=torch.float32) - 0.5) * (2 * noise)
            img = img + nz.to(dev, dt)
        img = torch.clamp(torch.round(img), 0, 255).to(torch.float32)
        images.append(img)
        Ks.append(K.astype(np.float32))
        Rs.append(Rm.reshape(9).astype(np.float32))
        ts.append((-Rm @ c).astype(np.float32))
    sc = Scene(width, height, num_src, images, Ks, Rs, ts, 1.0, 4.0, gt_depth, view_depths)
    sc.view_normals, sc.view_textured = view_normals, view_textured   # with keep_view_depths: analytic normal and "has texture" mask of every view
    return sc
