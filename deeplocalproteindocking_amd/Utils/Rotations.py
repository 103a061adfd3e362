"""SOI rotation sets: (phi, theta, psi) rows -> R = Rz(psi) Rx(theta) Rz(phi), (N,3,3) float64.

Drop-in for /root/reference/src/Utils/Rotations.py:8-66 (class Rotations, attribute ``R``), as a
superset:
  * the file is found whether it is named ``oim6.eul`` (what Rotations.py:39 builds) or
    ``oim06.eul`` (what the reference actually ships in data/), so angle_inc 6 and 8 load;
  * N comes from the file's line count, not from the table at Rotations.py:42-55;
  * the matrix fill is vectorised (the reference writes 9 tensor elements per row from Python);
  * when no file exists for ``angle_inc`` the loader raises "Can't find rotation angles" exactly
    like Rotations.py:41,55 -- UNLESS the caller opts in (``allow_generated=True``, or
    ``DLPD_ALLOW_GENERATED_ROTATIONS=1`` in the environment): then a deterministic substitute set
    is generated with the same structure and size as the SOI files (round(41253/inc^2)
    Fibonacci-sphere directions (theta, psi) x (360/inc) uniform in-plane angles phi; e.g. 4
    degrees, whose data/oim04.eul is absent from the reference, .MISSING_LARGE_BLOBS:1), a
    warning is issued and ``self.source`` is "generated".  Poses of a generated set are not
    comparable with the reference's rotation indices, hence never silently.
The SOI data files carry MitchellLab's licence (reference README.md:63) and are not
redistributed here: point ``DLPD_ROTATIONS_DIR`` at a directory holding them.
"""
import math
import os
import warnings

import numpy as np
import torch

try:  # `src` is this package when the reference's drivers run on top of it
    from src import REPOSITORY_DIR
except Exception:  # pragma: no cover - plain package import
    from deeplocalproteindocking_amd import REPOSITORY_DIR


def euler_to_matrices(phi, theta, psi):
    """Rotations.writeMatrix (Rotations.py:14-32) for arrays of angles, float64."""
    phi, theta, psi = (np.asarray(a, dtype=np.float64) for a in (phi, theta, psi))
    cpsi, spsi = np.cos(psi), np.sin(psi)
    cth, sth = np.cos(theta), np.sin(theta)
    cphi, sphi = np.cos(phi), np.sin(phi)
    R = np.empty(phi.shape + (3, 3), dtype=np.float64)
    R[..., 0, 0] = cpsi * cphi - spsi * cth * sphi
    R[..., 0, 1] = -cpsi * sphi - spsi * cth * cphi
    R[..., 0, 2] = spsi * sth
    R[..., 1, 0] = spsi * cphi + cpsi * cth * sphi
    R[..., 1, 1] = -spsi * sphi + cpsi * cth * cphi
    R[..., 1, 2] = -cpsi * sth
    R[..., 2, 0] = sth * sphi
    R[..., 2, 1] = sth * cphi
    R[..., 2, 2] = cth
    return R


def generated_set_size(angle_inc):
    nphi = int(round(360.0 / float(angle_inc)))
    nsphere = int(round(41252.96 / float(angle_inc) ** 2))
    return nsphere, nphi


def generate_angles(angle_inc):
    """Substitute SOI-like set: Fibonacci sphere x uniform phi, rows ordered like the SOI files
    (all phi of one direction consecutively, phi ascending from -pi)."""
    nsphere, nphi = generated_set_size(angle_inc)
    i = np.arange(nsphere, dtype=np.float64)
    theta = np.arccos(1.0 - 2.0 * (i + 0.5) / nsphere)
    golden = math.pi * (3.0 - math.sqrt(5.0))
    psi = np.mod(i * golden + math.pi, 2.0 * math.pi) - math.pi
    phi = -math.pi + 2.0 * math.pi * np.arange(nphi, dtype=np.float64) / nphi
    ang = np.empty((nsphere, nphi, 3), dtype=np.float64)
    ang[:, :, 0] = phi[None, :]
    ang[:, :, 1] = theta[:, None]
    ang[:, :, 2] = psi[:, None]
    return ang.reshape(-1, 3)


def find_rotation_file(angle_inc):
    names = []
    if float(angle_inc) == int(angle_inc):
        names = ["oim%d.eul" % int(angle_inc), "oim%02d.eul" % int(angle_inc)]
    dirs = []
    if os.environ.get("DLPD_ROTATIONS_DIR"):
        dirs.append(os.environ["DLPD_ROTATIONS_DIR"])
    dirs += [os.path.join(REPOSITORY_DIR, "data")]
    for d in dirs:
        for n in names:
            f = os.path.join(d, n)
            if os.path.exists(f):
                return f
    return None


class Rotations(object):
    def __init__(self, angle_inc=12, allow_generated=None, verbose=True):
        self.angle_inc = angle_inc
        if allow_generated is None:
            allow_generated = os.environ.get("DLPD_ALLOW_GENERATED_ROTATIONS", "") not in ("", "0")
        self.loadSOI(angle_inc, allow_generated)
        if verbose:
            print("Angle increment:", angle_inc)
            print("Number of rotations:", self.R.size(0), "(%s)" % self.source)

    def loadSOI(self, angle_inc, allow_generated=False):
        filename = find_rotation_file(angle_inc)
        if filename is not None:
            ang = np.loadtxt(filename, dtype=np.float64).reshape(-1, 3)
            self.source = filename
        elif allow_generated:
            ang = generate_angles(angle_inc)
            self.source = "generated"
            warnings.warn("dlpd: no SOI rotation file for angle_inc=%s (DLPD_ROTATIONS_DIR / data/): using a GENERATED "
                          "substitute set of %d rotations -- rotation indices are not the reference's" %
                          (angle_inc, ang.shape[0]))
        else:
            raise Exception("Can't find rotation angles:", angle_inc)
        self.angles = ang
        self.R = torch.from_numpy(euler_to_matrices(ang[:, 0], ang[:, 1], ang[:, 2]))
