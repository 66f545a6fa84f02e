"""Synthetic workloads of BASELINE.json / SURVEY.md 8(d) built with the product's own host + device code
(no oracle involved): model, orientation set, CTF grid, particle stack rendered by the engine itself."""
import math

import numpy as np

from . import hostlib
from .engine import POINT_DTYPE, Engine, ParamDevice

ELECWAVEL = np.float32(0.019866)  # reference default, param.cpp:86


def synth_model(n=2000, sigma=25.0, rmax=60.0, seed=20260101):
    """2 000 pseudo-C-alpha points ~ N(0, sigma) truncated to r < rmax, residue-like radii/electron counts."""
    rng = np.random.default_rng(seed)
    pts = np.zeros(n, dtype=POINT_DTYPE)
    k = 0
    while k < n:
        p = rng.normal(0.0, sigma, size=3)
        if np.linalg.norm(p) < rmax:
            pts["pos"][k] = p
            k += 1
    pts["radius"] = rng.uniform(2.25, 3.4, size=n)
    pts["density"] = rng.uniform(40.0, 108.0, size=n)
    return pts


def random_quaternions(n, seed=20260103):
    """Uniform random unit quaternions (the reference's QUATERNION_LIST files cannot travel to the GPU box)."""
    rng = np.random.default_rng(seed)
    q = rng.normal(size=(n, 4))
    q /= np.linalg.norm(q, axis=1)[:, None]
    return q.astype(np.float32)


def make_param_device(N, maxD, grid, nOrient, ctf_steps, nAmp, px, priorMod=1.0):
    pd = ParamDevice()
    pd.maxDisplaceCenter = maxD
    pd.GridSpaceCenter = grid
    pd.NumberPixels = N
    pd.NumberFFTPixels1D = N // 2 + 1
    pd.NxDisp = 2 * (maxD // grid) + 1
    pd.NtotDisp = pd.NxDisp ** 2
    pd.Ntotpi = float(N * N)
    pd.sigmaPriorbctf = 100.0
    fac = math.pi * 2.0 * 10000 * float(ELECWAVEL)
    pd.sigmaPriordefo = np.float32(2.0 * fac)
    pd.Priordefcent = np.float32(3.0 * fac)
    pd.sigmaPrioramp = 0.5
    pd.Priorampcent = 0.0
    pd.writeAngles = 0
    pd.tousepsf = 0
    voluang = np.float32(1.0 / nOrient * priorMod)
    pd.volu = hostlib.volume_element(voluang, grid, maxD, px, nAmp, ctf_steps[2], ctf_steps[1], pd.sigmaPriorbctf,
                                     pd.sigmaPriordefo, pd.sigmaPrioramp)
    return pd


class Workload:
    """config-2 shaped workload: N^2 maps, nP particles, nOrient orientations, CTF grid, +-maxD displacement."""

    def __init__(self, N=224, nP=1000, nOrient=4608, nEnv=5, nDefocus=1, maxD=10, grid=1, px=1.77, npts=2000,
                 snr=0.05, device=0, algo=1, orient_seed=20260103, render=True, write_angles=0, blocks=1, block=0):
        """nOrient orientations PER BLOCK; the global list is `blocks` such blocks (block b seeded orient_seed + b), this
        engine is the shard that owns block `block` (the reference's MPI blocks).  Particles are rendered from block 0,
        so every shard of a job sees the same particle stack as the one-block job."""
        self.N, self.nP, self.nOrient, self.px = N, nP, nOrient, np.float32(px)
        self.blocks, self.block = blocks, block
        self.o0, self.o1 = block * nOrient, (block + 1) * nOrient
        fac = math.pi * 2.0 * 10000 * float(ELECWAVEL)
        amp = (np.float32(0.1), np.float32(0.1), 1)
        phase = (np.float32(1.0 * fac), np.float32(4.0 * fac), nDefocus)
        env = (np.float32(2.0), np.float32(300.0), nEnv)
        self.refCTF, self.ctfParam, self.steps = hostlib.ctf_kernels(N, self.px, amp, phase, env)
        self.nCTF = len(self.ctfParam)
        self.pd = make_param_device(N, maxD, grid, nOrient * blocks, self.steps, 1, self.px)
        self.pd.writeAngles = int(write_angles)
        scale = N * px / (224 * 1.77)
        pts = synth_model(npts, 25.0 * scale, 60.0 * scale)
        self.points, self.NormDen = hostlib.center_model(pts)
        self.angles = np.concatenate([random_quaternions(nOrient, orient_seed + b) for b in range(blocks)])
        if blocks == 1 and not write_angles:
            self.engine = Engine(self.pd, nP, nOrient, self.nCTF, algo=algo, device=device)
        else:  # shard of the global list: angle entries of the own block only, kept on the device
            self.engine = Engine(self.pd, nP, nOrient * blocks, self.nCTF, algo=algo, device=device,
                                 shard=(self.o0, self.o1))
        E = self.engine
        E.upload_ctf(self.refCTF, self.ctfParam)
        E.upload_model(self.points, self.NormDen, self.px)
        E.upload_orientations(self.angles, True)
        self.maps = None
        if render:
            self.maps = self.render_particles(snr)
            E.upload_particle_maps(self.maps)

    def render_particles(self, snr, seed=20260102, maxshift=8):
        """Particle p: orientation (7919 p) mod nOrient, CTF p mod nCTF, integer shift in [-maxshift, maxshift]^2,
        unit-variance signal * sqrt(snr) + N(0,1) noise, z-scored (as the MRC reader does, map.cpp:831-845)."""
        N = self.N
        maps = np.zeros((self.nP, N, N), dtype=np.float32)
        clean = {}  # unit-variance noise-free image per (orientation, CTF); large stacks revisit the pairs
        for p in range(self.nP):
            rng = np.random.default_rng(seed + p)
            o = (7919 * p) % self.nOrient
            c = p % self.nCTF
            if (o, c) not in clean:
                spec, _, _ = self.engine.debug_convolution(o, c)
                z = spec[..., 0] + 1j * spec[..., 1]
                img = np.fft.irfft2(z, s=(N, N))
                sd = img.std()
                img = (img - img.mean()) / (sd if sd > 0 else 1.0)
                if self.nP > self.nOrient * self.nCTF:
                    clean[(o, c)] = img
            else:
                img = clean[(o, c)]
            sx, sy = rng.integers(-maxshift, maxshift + 1, size=2)
            img = np.roll(img, (int(sx), int(sy)), axis=(0, 1))
            img = img * math.sqrt(snr) + rng.normal(size=(N, N))
            img = (img - img.mean()) / img.std()
            maps[p] = img.astype(np.float32)
        return maps

    @property
    def comparisons_per_pass(self):
        return self.nOrient * self.nCTF * self.nP
