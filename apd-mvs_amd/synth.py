"""Deterministic synthetic multi-view scenes (stand-ins for ETH3D / T&T, which are not in the image).

SURVEY.md section 8(d): analytic scene rendered by exact ray/plane intersection, pinhole cameras
`K = [f 0 cx; 0 f cy; 0 0 1]`, `f = 0.9 W`, procedural texture (sum of four sinusoid products) with
optional textureless rectangles + per-view noise, quantised to 8 bit like a decoded JPEG
(APD.cpp:410-413 reads uint8 and converts to float).  Works on CPU or GPU torch tensors.
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


def make_scene(width, height, num_src, seed=0, textureless=0.0, rotate=True, baseline=0.06,
               ref_view=0, device="cpu", noise=1.5, keep_view_depths=False):
    """Render view `ref_view` of a ring of cameras as the reference and `num_src` neighbours.

    textureless: fraction (0..~0.4) of the surface covered by low-texture rectangles.
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
        rots.append(_lookat_rotation(c, target) if (rotate and not (k == 0 and ref_view == 0)) else np.eye(3))
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

    dev = torch.device(device)
    dt = torch.float64
    ys, xs = torch.meshgrid(torch.arange(height, device=dev, dtype=dt), torch.arange(width, device=dev, dtype=dt), indexing="ij")
    images, Ks, Rs, ts = [], [], [], []
    gt_depth = None
    view_depths = [] if keep_view_depths else None
    for k in range(num_src + 1):
        Rm, c = rots[k], centers[k]
        Rt = torch.tensor(Rm.T, device=dev, dtype=dt)
        cc = torch.tensor(c, device=dev, dtype=dt)
        dcx, dcy = (xs - cx) / f, (ys - cy) / f
        # ray direction in world = R^T [dcx, dcy, 1]
        dw = [Rt[i, 0] * dcx + Rt[i, 1] * dcy + Rt[i, 2] for i in range(3)]
        best_s = None
        for n, d in planes:
            num = -(float(n @ c) + d)
            den = n[0] * dw[0] + n[1] * dw[1] + n[2] * dw[2]
            s = num / den
            s = torch.where(s > 1e-6, s, torch.full_like(s, 1e9))
            best_s = s if best_s is None else torch.minimum(best_s, s)
        P = [cc[i] + best_s * dw[i] for i in range(3)]
        if k == 0:
            gt_depth = best_s.to(torch.float32)  # camera-frame z: ray has z=1 in camera coordinates
        if keep_view_depths:
            view_depths.append(best_s.to(torch.float32))
        X, Y = P[0], P[1]
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
        img = 128.0 + tex * scale
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
    return Scene(width, height, num_src, images, Ks, Rs, ts, 1.0, 4.0, gt_depth, view_depths)
