"""Exhaustive rotation x translation search driver -- drop-in for the reference's ``Docker``
class (/root/reference/src/Docker/Docker.py:17-238): same constructor, ``new_log``, ``cleanup``,
``update_top``, ``write_conformations``, ``load_batch``, ``dockE3``, ``dockSE3`` and attributes
(``top_list``, ``rot.R``, ``box_size``, ``resolution``, ``box_length``, ``max_conf``, ``log``).

It is constructed and driven exactly as /root/reference/src/local_test.py:55,65-71 does
(``Docker(docking_model=..., angle_inc=..., box_size=80, resolution=1.25, max_conf=2000,
randomize_rot=True)``, ``new_log``, ``dockSE3(rec, lig, batch_size=2)``); every further keyword is
optional.

What differs underneath:
  * the hot loop (Docker.py:211-236) runs in libdlpd.so with no host synchronisation; the clash
    mask, the filter MLP and the mask multiply are fused into the inverse FFT, and the top list
    lives on the device until the pair is finished;
  * the caller's ``batch_size`` does NOT size the launches: the ranked list is independent of how
    the rotations are batched (merge key (score, rotation, pick); tested), so ``dockSE3`` /
    ``dockE3`` always score ``launch_batch`` (32) rotations per launch -- the receptor slab in K2 is
    amortised over the launch batch, and ``batch_size=2`` of local_test.py:69,71 would cost 8x the
    launches;
  * the scoring call ``docking_model(receptor_volumes, ligand_volumes_rotated)`` (Docker.py:229) is
    replaced by the fused kernels only when that is what the model computes -- the reference's
    ``GlobalDockingModel.forward`` with an MLP ``SimpleFilter`` (``Models.fused_filter_parameters``).
    Any other model / filter module is CALLED, on volumes rotated and correlated by the stand-alone
    HIP ops, followed by the device top-K (``_dock_volumes_generic``);
  * atoms come from ``Utils.FullAtom.CoordsBackend`` (PDB reader, 11-type typing, GPU density
    projection; TorchProteinLibrary itself is absent, SURVEY.md 8f rows 1,3), created on first use;
    ``coords_backend=`` swaps in another implementation;
  * ``dock_volumes`` is the volume-level entry (a superset): it takes representation volumes
    directly, which is what the synthetic BASELINE configs use;
  * rotations can be sharded over ranks (``rank``/``world_size``): rank r scores rotations
    r, r+W, r+2W, ... and one all-gather of the per-rank top lists + a deterministic merge by
    (score, rotation, pick) reproduces the single-process list.
"""
import atexit
import os

import numpy as np
import torch

from deeplocalproteindocking_amd.engine import DeviceTopList, DockingEngine
from deeplocalproteindocking_amd._lib import get_lib
from deeplocalproteindocking_amd.Utils.Conventions import VolumeConventions
from deeplocalproteindocking_amd.Utils.Rotations import Rotations, euler_to_matrices


def fused_filter_parameters(model):
    from deeplocalproteindocking_amd.Models.DockingModels import fused_filter_parameters as f
    return f(model)


class _PivotRotation(object):
    """ops.VolumeRotation per grid size, with the Docker's rotation conventions (pivot, scale, axis order)."""

    def __init__(self, docker):
        self.docker, self.ops = docker, {}

    def __call__(self, volume, R):
        from deeplocalproteindocking_amd.ops import VolumeRotation
        L = volume.shape[-1]
        if L not in self.ops:
            cv = self.docker.conventions
            self.ops[L] = VolumeRotation(center=self.docker.rotation_pivot(L), lib=self.docker._lib,
                                         scale=cv.rotation_scale, axis_order=cv.rotation_axis_order,
                                         transpose=cv.rotation_transpose)
        return self.ops[L](volume, R)


class _FixedRotations(object):
    def __init__(self, R):
        self.R = torch.as_tensor(np.asarray(R, dtype=np.float64)).reshape(-1, 3, 3)
        self.source = "explicit"


def random_rotation(generator=None, seed=None):
    """Uniform random rotation (1,3,3) float64 -- stand-in for TPL getRandomRotation(1)
    (Docker.py:44); the TPL RNG stream cannot be matched (source absent).
    Without ``generator`` the numbers come from a generator of their own -- seeded with ``seed`` or, for ``None``, from
    the environment variable DLPD_ROTATION_SEED or, for ``None`` and no such variable, from the operating system's
    entropy -- so that the rotation neither repeats from run to run (torch's global generator starts from a fixed default
    seed) nor depends on what else has drawn from the process-global stream."""
    if generator is None:
        if seed is None and os.environ.get("DLPD_ROTATION_SEED", "") != "":
            seed = int(os.environ["DLPD_ROTATION_SEED"])   # an UNCHANGED caller (local_test.py passes no such keyword)
        generator = torch.Generator()
        generator.manual_seed(int(seed)) if seed is not None else generator.seed()
    u = torch.rand(3, generator=generator, dtype=torch.float64).numpy()
    phi, psi = 2 * np.pi * u[0] - np.pi, 2 * np.pi * u[2] - np.pi
    theta = np.arccos(1.0 - 2.0 * u[1])
    return torch.from_numpy(euler_to_matrices(phi, theta, psi)).reshape(1, 3, 3)


def all_gather_top_entries(entries, K, world_size, process_group=None, device="cpu", always=False):
    """The ONE collective of the rotation-sharded search: every rank contributes its local top list
    ``entries`` = (rot, flat index, score, pick) as a fixed-size block, one ``all_gather`` (RCCL over xGMI on the
    ``nccl`` backend, gloo on CPU), then the same deterministic merge on every rank -- sort by (score, rotation, pick),
    keep K -- which reproduces the single-process list exactly: the global top-K is a subset of the union of the
    per-shard top-Ks and the key is the reference's stable insertion order (Docker.py:100-105).
    Used by ``Docker`` and by ``bench.py``'s multi-rank leg alike.  ``always`` runs the collective for a group of one
    too (the one-GPU boxes' RCCL check, scripts/rccl_one_rank_check.py)."""
    if world_size <= 1 and not always:
        return entries
    import torch.distributed as dist
    pack = torch.zeros(4, K + 1, dtype=torch.float64)
    n = len(entries[0])
    pack[0, 0] = n
    pack[0, 1:1 + n] = torch.from_numpy(np.asarray(entries[0], dtype=np.float64))
    pack[1, 1:1 + n] = torch.from_numpy(np.asarray(entries[1], dtype=np.float64))
    score_bits = np.ascontiguousarray(entries[2], dtype=np.float32).view(np.uint32)
    pack[2, 1:1 + n] = torch.from_numpy(score_bits.astype(np.float64))   # exact score bits
    pack[3, 1:1 + n] = torch.from_numpy(np.asarray(entries[3], dtype=np.float64))
    backend = dist.get_backend(process_group)
    buf = pack.to(device) if backend == "nccl" else pack
    out = [torch.empty_like(buf) for _ in range(world_size)]
    dist.all_gather(out, buf, group=process_group)
    parts = []
    for o in out:
        o = o.cpu().numpy()
        m = int(o[0, 0])
        parts.append((o[0, 1:1 + m].astype(np.int64), o[1, 1:1 + m].astype(np.int64),
                      o[2, 1:1 + m].astype(np.uint32).view(np.float32), o[3, 1:1 + m].astype(np.int64)))
    return DeviceTopList.merge_entries(parts, K)


LAUNCH_BATCH = 32      # rotations per launch of the fused pipeline (16 through round 5: EXPERIMENTS.md R6)


class PreparedPair(object):
    """Everything ``dockSE3`` / ``dockE3`` do for a target BEFORE the rotation loop (Docker.py:184-209 / :135-161): the
    two PDB files parsed, typed and centred, the receptor projected and represented, its spectrum in an engine, the
    ligand's volumes (SE3) and its atoms on the device.  ``Docker.prepare`` builds one -- from a host thread of its own if
    the caller wants the next target prepared while the current one is searched (local_test.py sweep), but on the CALLER'S
    stream: device work of a preparation is never put beside a running search (see ``Docker.prepare``) -- and the dock
    call consumes it.  ``ready`` is only set by the diagnostic ``stream=`` form and then orders the consumer's stream
    after the producer's."""

    def __init__(self, group, ureceptor, uligand, slot):
        self.group, self.ureceptor, self.uligand, self.slot = group, ureceptor, uligand, slot
        self.receptor = self.receptor_volumes = self.receptor_forbidden = self.ligand_volumes = None
        self.ligand_atoms = None           # (coords, num_atoms_of_type, offsets) on the device, origin-centred
        self.engine = None                 # fused engine with receptor (and, SE3, ligand) already set; None: other paths
        self.ready = None                  # torch.cuda.Event recorded on the producer's stream
        self.seconds = 0.0                 # host time of the preparation

    def wait(self, device):
        """The caller's current stream waits for the preparation; tensors made on the producer's stream are marked as
        used here so that the caching allocator does not hand their memory out while this stream still reads it."""
        if self.ready is None or device.type != "cuda":
            return
        cur = torch.cuda.current_stream(device)
        cur.wait_event(self.ready)
        for t in self.tensors():
            t.record_stream(cur)

    def tensors(self):
        out = [self.receptor, self.receptor_forbidden] + list(self.receptor_volumes or []) + list(self.ligand_volumes or []) + \
            list(self.ligand_atoms or [])
        return [t for t in out if torch.is_tensor(t) and t.is_cuda]


class Docker:
    def __init__(self, docking_model, angle_inc=15.0, box_size=80, resolution=1.25, max_conf=1000,
                 randomize_rot=False, rotations=None, device="cuda", coords_backend=None,
                 rank=0, world_size=1, process_group=None, lib=None, launch_batch=None, rotation_center=None,
                 conventions=None, rotation_seed=None, collectives_with_one_rank=False):
        self.docking_model = docking_model
        self.log = None

        self.box_size = box_size
        self.resolution = resolution
        self.box_length = box_size * resolution

        self.max_conf = max_conf
        self.rot = Rotations(angle_inc=angle_inc) if rotations is None else _FixedRotations(rotations)

        self.box_center = torch.zeros(1, 3, dtype=torch.double, device='cpu')
        self.box_center.fill_(self.box_length / 2.0)

        self.device = torch.device(device)
        self.coords_backend = coords_backend
        self.rank, self.world_size, self.process_group = int(rank), int(world_size), process_group

        # Docker.py:42-45.  Rotation-sharded over ranks, every rank must score the SAME rotated receptor: rank 0's
        # matrix is broadcast (here when the process group already exists, else at the start of the first dock* call);
        # rotation_seed makes it reproducible (None: a fresh one per Docker, as "random" says).
        # collectives_with_one_rank: run the broadcasts and the all-gather for a group of ONE rank too -- how a one-GPU box
        # takes this class through RCCL itself (local_test.py -force_group 1); the results are the same by construction
        self.collectives_with_one_rank = bool(collectives_with_one_rank)
        self.randomize_rot = randomize_rot
        self._randR_shared = self.world_size <= 1 and not self.collectives_with_one_rank
        if self.randomize_rot:
            self.randR = random_rotation(seed=rotation_seed)
            self._share_random_rotation(required=False)
            if self.rank == 0 or not self._randR_shared:
                print("Adding random rotation to the receptor:", self.randR)
        self._lib = lib
        self.launch_batch = int(launch_batch or os.environ.get("DLPD_LAUNCH_BATCH", LAUNCH_BATCH))
        # Pivot of the trilinear VOLUME rotation in voxel-index units (build-defined: TorchProteinLibrary's
        # VolumeRotation has no source here).  None: index L/2 of each grid (the box centre when voxel i spans
        # [i, i+1) * resolution, Docker.py:221-223); "grid_sample": (L-1)/2 of each grid (the centre of
        # torch's align_corners=False sampling grid); a number: that index on the fine grid, scaled to coarser ones.
        # conventions: a Utils.Conventions.VolumeConventions (or the path of the JSON scripts/calibrate_tpl.py writes): the
        # remaining build-defined choices -- rotation scale and axis order, what VolumeConvolution(clip) clamps, the
        # density splat -- next to the pivot.  ``rotation_center`` given explicitly wins over the file's.
        if isinstance(conventions, str):
            conventions = VolumeConventions.load(conventions)
        # (a copy: the rotation_center setter below must not reach into an object the caller shares between Dockers)
        self.conventions = conventions.copy() if conventions is not None else VolumeConventions()
        if rotation_center is not None:
            self.conventions = VolumeConventions.from_dict(dict(self.conventions.to_dict(), rotation_center=rotation_center))
        self._ops_cache = {}
        # box sizes without a compiled plan: on the fused kernels inside the next compiled box (_dock_volumes_embedded);
        # False: the plan-free stand-alone ops (ops.VolumeConvolution._forward_generic), any box up to 128
        self.embed_uncompiled_boxes = True
        # dockE3 on the fused engine: a plugin with occupancy maps does not write the tiles it skips (the engine goes by the
        # maps); False: every voxel of the batch's volumes is written (same lists; DLPD_UNWRITTEN_ACTIVATIONS=0)
        self.unwritten_activations = os.environ.get("DLPD_UNWRITTEN_ACTIVATIONS", "1") != "0"
        # the search side's occupancy maps (DockingEngine(sparse_k1=)): None = decided per ligand; DLPD_K1_OCCUPANCY=0 / 1 forces
        # the dense / the map kernels for whole programs (A/B runs: the lists are the same either way)
        self.k1_occupancy = {"0": False, "1": True}.get(os.environ.get("DLPD_K1_OCCUPANCY", ""))
        self._top = None            # DeviceTopList behind update_top()
        self.top_list = []
        self.engine = None
        self._engine_pool = {}      # slot -> (key, engine): slot 1 exists only while targets are prepared ahead (prepare)
        atexit.register(self.cleanup)

    def _share_random_rotation(self, required=True):
        """world_size > 1: replace this rank's ``randR`` by rank 0's (one broadcast of nine doubles, once per Docker).
        Without it the ranks would score differently rotated receptors and merge them into one silently wrong list."""
        if not self.randomize_rot or self._randR_shared:
            return
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()):
            if required:
                raise Exception("Docker(world_size=%d, randomize_rot=True) needs an initialised torch.distributed "
                                "process group to share the random rotation" % self.world_size)
            return
        group = self.process_group
        src = dist.get_global_rank(group, 0) if group is not None else 0
        buf = self.randR.to(dtype=torch.float64).reshape(9).clone()
        if dist.get_backend(group) == "nccl":
            buf = buf.to(self.device)
        dist.broadcast(buf, src=src, group=group)
        self.randR = buf.cpu().reshape(1, 3, 3)
        self._randR_shared = True

    # ------------------------------------------------------------------ the reference's operator objects
    # Docker.py:29-40 creates TorchProteinLibrary operators as attributes; nothing outside the class reads them, but
    # a subclass may.  They are thin views of the coords backend / the volume ops with TPL's call signatures,
    # created on first use (the constructor must not need a GPU or a backend).
    @property
    def rotate(self):                       # CoordsRotate()(coords, R, num_atoms)
        return self._need_backend().rotate

    @property
    def translate(self):                    # CoordsTranslate()(coords, T, num_atoms)
        return self._need_backend().translate

    translation = translate

    @property
    def pdb2coords(self):                   # PDB2CoordsUnordered()(filenames)
        return self._need_backend().pdb2coords

    @property
    def assignTypes(self):                  # Coords2TypedCoords()(coords, resnames, atomnames, num_atoms)
        return self._need_backend().assign_types

    @property
    def project(self):                      # TypedCoords2Volume(box_size, resolution)(coords, num_atoms_of_type, offsets)
        be = self._need_backend()
        return lambda coords, num_atoms_of_type, offsets: be.project(coords, num_atoms_of_type, offsets, self.box_size,
                                                                     self.resolution, self.device)

    @property
    def convolve(self):                     # VolumeConvolution()(volume1, volume2): no clip (Docker.py:32)
        from deeplocalproteindocking_amd.ops import VolumeConvolution
        key = ("convolve", self.embed_uncompiled_boxes)
        if key not in self._ops_cache:      # one object per Docker, as the reference's attribute is
            self._ops_cache[key] = VolumeConvolution(lib=self._lib, embed=self.embed_uncompiled_boxes)
        return self._ops_cache[key]

    @property
    def vol_rotate(self):                   # VolumeRotation()(volume, R), with this Docker's rotation conventions
        if "vol_rotate" not in self._ops_cache:
            self._ops_cache["vol_rotate"] = _PivotRotation(self)
        return self._ops_cache["vol_rotate"]

    # ------------------------------------------------------------------ logging (Docker.py:63-84)
    def new_log(self, log_file_name, rewrite=True):
        """Open the .dat of the next target.  Resume rule of Docker.py:63-79: with ``rewrite=False`` a file
        that already holds more than one non-blank line counts as finished -- no log is opened and False is
        returned so the caller skips the target; anything else is (re)started from scratch."""
        self.cleanup()
        if not rewrite and self.log_is_complete(log_file_name):
            return False
        self.log = open(log_file_name, "w")
        return True

    @staticmethod
    def log_is_complete(log_file_name):
        """The test of the resume rule alone (Docker.py:66-76), without opening anything: does the file exist and hold
        more than one non-blank line?  (A sweep uses it to decide which target to prepare ahead.)"""
        if not os.path.exists(log_file_name):
            return False
        with open(log_file_name) as existing:
            return sum(1 for line in existing if line.split()) > 1

    def cleanup(self):
        """Close the current log, if any (also registered with atexit, Docker.py:46,81-84)."""
        log, self.log = self.log, None
        if log is not None:
            log.close()

    # ------------------------------------------------------------------ top list (Docker.py:86-105)
    def _library(self):
        if self._lib is None:
            self._lib = get_lib()
        return self._lib

    def update_top(self, V, rotation_index):
        """V (N,N,N) float32 device tensor.  Same observable behaviour as Docker.py:86-105: the
        max_conf picks are appended to ``self.top_list`` (stable sort, truncate) and the picked
        voxels of V are set to 0.0 in place.  The list is kept on the device; ``self.top_list``
        is refreshed from it."""
        if self._top is None or self._top.K != self.max_conf or self._top.device != V.device:
            self._top = DeviceTopList(self.max_conf, 1, V.device, self._library())
            self._top.reset()
        elif len(self.top_list) == 0:
            self._top.reset()                      # caller cleared top_list: a new pair starts
        Vc = V.contiguous()
        _, idx = self._top.select(Vc.reshape(1, -1), 1)
        self._top.merge(torch.tensor([rotation_index], dtype=torch.int32, device=V.device), 1)
        Vc.view(-1)[idx[0].long()] = 0.0           # Docker.py:98, V is mutated in place
        if Vc.data_ptr() != V.data_ptr():
            V.copy_(Vc)
        self.top_list = DeviceTopList.to_top_list(self._top.entries(), V.shape[0])

    # ------------------------------------------------------------------ output (Docker.py:107-133)
    def write_conformations(self):
        """One line per pose: the 9 entries of R row-major, the translation in Angstrom, the score -- 13
        tab-separated ``%f`` columns (Docker.py:107-133, byte-exact against fixture G4).  Grid indices at
        or beyond ``box_size`` are negative shifts (index - 2L); with ``randomize_rot`` both R and t are
        taken back to the receptor's original frame by randR^T."""
        if self.log is None:
            return
        L, N = self.box_size, 2 * self.box_size
        undo = self.randR.squeeze().t().to(torch.double) if self.randomize_rot else None
        rows = []
        for rot_index, x, y, z, score in self.top_list:
            r = self.rot.R[rot_index].to(torch.double)
            shift = torch.tensor([v - N if v >= L else v for v in (x, y, z)], dtype=torch.double)
            t = shift * self.resolution
            if undo is not None:
                t, r = undo @ t, undo @ r
            cols = [float(v) for v in r.reshape(-1)] + [float(v) for v in t] + [score]
            rows.append("\t".join("%f" % v for v in cols))
        if rows:
            self.log.write("\n".join(rows) + "\n")
        self.log.flush()

    @property
    def rotation_center(self):
        return self.conventions.rotation_center

    @rotation_center.setter
    def rotation_center(self, value):
        self.conventions.rotation_center = value
        self._ops_cache.pop("vol_rotate", None)          # (its per-size operators carry the pivot)

    def rotation_pivot(self, L):
        """Pivot index of the volume rotation on a grid of L voxels per edge (see ``rotation_center``)."""
        return self.conventions.pivot(L, self.box_size)

    def release_engine(self):
        """Drop the cached fused engine and its device workspaces (several GB at box 80: wsB for 32 rotations,
        double-buffered score volumes).  The next ``dock*`` call builds a new one."""
        for _, eng in self._engine_pool.values():
            eng.finish()
        self._engine_pool = {}
        self.engine, self._engine_key = None, None
        self._top = None

    # ------------------------------------------------------------------ the search on volumes
    def shard(self, nrot):
        """Rotation indices of this rank: interleaved, ascending."""
        return np.arange(self.rank, nrot, self.world_size, dtype=np.int64)

    def dock_volumes(self, receptor_volumes, ligand_volumes, receptor_forbidden=None, ligand_forbidden=None,
                     batch_size=None, rot_indices=None, write=True, clash_provider=None, model_batch=None, prepared=None):
        """Search all rotations for one pair given its representation volumes.

        receptor_volumes / ligand_volumes: lists of (1,C_i,L_i,L_i,L_i) (or (C_i,L_i,..)) tensors
        as returned by ``docking_model.representation``; *_forbidden: (L,L,L)-shaped clash
        densities (``None`` -> no clash exclusion).  The ligand forbidden volume is rotated with
        the same trilinear kernel as the representation unless ``clash_provider`` re-projects it
        from rotated atoms (Docker.py:221-224).
        batch_size: rotations per launch (None -> ``self.launch_batch``); the list does not depend
        on it.  model_batch: rotations per call of a model that has to be CALLED (generic path)."""
        model = self.docking_model
        model.eval() if hasattr(model, "eval") else None
        self._share_random_rotation()           # (write_conformations undoes randR: the same matrix on every rank)
        rec = [torch.as_tensor(v, dtype=torch.float32) for v in receptor_volumes]
        lig = [torch.as_tensor(v, dtype=torch.float32) for v in ligand_volumes]
        rec = [v.reshape((-1,) + tuple(v.shape[-3:])) for v in rec]
        lig = [v.reshape((-1,) + tuple(v.shape[-3:])) for v in lig]
        L = rec[0].shape[-1]
        if L != self.box_size:
            raise Exception("Volume size does not match box_size", L, self.box_size)
        nb = int(batch_size or self.launch_batch)
        R_all = self.rot.R
        ids = self.shard(R_all.shape[0]) if rot_indices is None else np.asarray(rot_indices, dtype=np.int64)
        params = fused_filter_parameters(model)
        # prepared: a PreparedPair whose engine already holds this receptor's spectrum and this ligand (Docker.prepare)
        pre_eng = prepared.engine if prepared is not None else None
        eng = pre_eng if pre_eng is not None else self._make_engine(rec, receptor_forbidden, nb, params)
        entries = None
        if eng is None and self.embed_uncompiled_boxes and not self._library().call("dlpd_grid_supported", int(L)):
            entries = self._dock_volumes_embedded(rec, lig, receptor_forbidden, ligand_forbidden, nb, ids,
                                                  clash_provider, params)
        if entries is not None:
            self.path = "embedded"
        else:
            self.path = "fused" if eng is not None else ("ops" if self._ops_path_ok(rec, params) else "call")
        if entries is not None:
            pass
        elif eng is not None:
            two_res = eng.C1 > 0
            if pre_eng is None:
                eng.set_ligand(lig[0], ligand_forbidden if ligand_forbidden is not None else torch.zeros(L, L, L),
                               lig[1] if two_res else None)
            eng.clash_provider = clash_provider
            eng.reset_top()
            eng.search(R_all[ids], rot_ids=ids)
            self.engine = eng
            entries = eng.top_entries()
        elif self.path == "ops":
            entries = self._dock_volumes_multires(rec, lig, receptor_forbidden, ligand_forbidden, nb, ids,
                                                  clash_provider, params)
        else:
            entries = self._dock_volumes_generic(rec, lig, receptor_forbidden, ligand_forbidden,
                                                 int(model_batch or nb), ids, clash_provider)
        entries = self._gather(entries)
        self.top_list = DeviceTopList.to_top_list(entries, 2 * L)
        if write:
            self.write_conformations()
        return self.top_list

    def _make_engine(self, rec, receptor_forbidden, batch_size, params, inner_box=None, slot=0):
        """DockingEngine for one receptor (one resolution, or the reference's [C0 @ L, C1 @ L/2] pair);
        None when the fused pipeline has no kernel for the shape (other grids or resolution layouts,
        hidden width above 32) or the model's scoring is not the MLP it fuses: the stand-alone-op paths
        take over."""
        if params is None:
            return None
        lib = self._library()
        L = rec[0].shape[-1]
        if not lib.call("dlpd_grid_supported", int(L)):
            return None
        two_res = len(rec) == 2 and rec[1].shape[-1] * 2 == L and bool(lib.call("dlpd_grid_supported", L // 2))
        if not (len(rec) == 1 or two_res):
            return None
        W1, b1, W2, b2 = params
        HP = int(lib.call("dlpd_fused_hidden_pad", int(W1.shape[0]), int(L), int(two_res)))
        if HP < 0 or W1.shape[1] != sum(v.shape[0] for v in rec):
            return None
        model = self.docking_model
        # one engine (multi-GB workspaces, side stream, top-list buffers) serves every pair of the same shape:
        # local_test.py docks hundreds of targets with one Docker
        C, C1, has_clash = rec[0].shape[0], (rec[1].shape[0] if two_res else 0), receptor_forbidden is not None
        # inner_box: the volumes are inner_box^3 boxes in the corner of the L^3 ones (_dock_volumes_embedded): pivots and
        # crop of the rotation are the small box's
        Lp = int(inner_box or L)
        cv = self.conventions
        if cv.clip_mode == "input" and Lp < L:
            return None          # clamped INPUTS are not combined with embedded boxes: the stand-alone ops take the pair
        key = (int(L), int(C), int(C1), has_clash, HP, int(self.max_conf),
               int(batch_size), str(self.device), self.rotation_pivot(Lp), Lp, cv.scale(Lp), cv.rotation_axis_order, cv.clip_mode,
               cv.rotation_transpose)
        pooled = self._engine_pool.get(slot)
        eng = pooled[1] if pooled is not None and pooled[0] == key else None
        if eng is None:
            eng = DockingEngine(L, C, W1.cpu(), b1.cpu(), W2.cpu(), b2.cpu(), clip=getattr(model, "clip", 5.0),
                                threshold_clash=model.threshold_clash, has_clash=has_clash, max_conf=self.max_conf,
                                batch=batch_size, device=self.device, lib=self._lib, coarse_channels=C1,
                                center=self.rotation_pivot(Lp), coarse_center=self.rotation_pivot(Lp // 2),
                                extent=(Lp if Lp < L else None), rotation_scale=cv.scale(Lp),
                                coarse_rotation_scale=cv.scale(Lp // 2), rotation_axis_order=cv.rotation_axis_order,
                                clip_mode=cv.clip_mode, rotation_transpose=cv.rotation_transpose, sparse_k1=self.k1_occupancy)
            self._engine_pool[slot] = (key, eng)
        else:
            eng.finish()
            eng.set_filter(W1.cpu(), b1.cpu(), W2.cpu(), b2.cpu())
            eng.clip, eng.threshold = getattr(model, "clip", 5.0), float(model.threshold_clash)
        eng.set_receptor(rec[0], receptor_forbidden, rec[1] if two_res else None)
        eng.holds_pair = None                            # (prepare() names the PreparedPair this content belongs to)
        if slot == 0:
            self.engine, self._engine_key = eng, key
        return eng

    @staticmethod
    def _embed(v, Le):
        """v (..., l, l, l) in the corner of a zero (..., Le, Le, Le) volume."""
        out = torch.zeros(tuple(v.shape[:-3]) + (Le, Le, Le), dtype=torch.float32, device=v.device)
        l = v.shape[-1]
        out[..., :l, :l, :l] = v
        return out

    def _embedding_box(self, L, two_res):
        """The smallest box WITH a compiled plan that holds an L^3 volume (and, for the reference's two
        resolutions, whose half holds the L/2 grid); None if there is none."""
        lib = self._library()
        for Lc in (32, 40, 64, 80):
            if Lc > L and lib.call("dlpd_grid_supported", Lc):
                if not two_res or (L % 2 == 0 and lib.call("dlpd_grid_supported", Lc // 2)):
                    return Lc
        return None

    def _dock_volumes_embedded(self, rec, lig, rec_forb, lig_forb, batch_size, ids, clash_provider, params):
        """A box size without a compiled plan on the FUSED kernels (box_size is a free argument of the reference,
        Docker.py:18,22-24,31): the L^3 volumes sit in the corner of the next compiled box Lc^3, zeros around them.
        The correlation of two L-sized volumes is linear for every translation |t| < L on any grid of at least 2L
        points, so the 2Lc grid holds the reference's (2L)^3 grid exactly: index t for 0 <= t <= L (t = L: no overlap,
        zero -- the reference's wrap plane) and 2Lc + t for -L < t < 0.  The engine rotates the embedded ligand about the
        SMALL box's pivot and crops the result to the small box (K1's ``extent``, include/dlpd.h), so a batch costs what
        it costs at box Lc; the scores of the reference's grid are gathered out of the larger one -- monotonic in every
        index, so ties keep the reference's order -- for the device top-K.  None when no compiled box fits or the model's
        scoring is not the fused MLP."""
        if params is None:
            return None
        L = rec[0].shape[-1]
        two_res = len(rec) == 2 and rec[1].shape[-1] * 2 == L
        if not (len(rec) == 1 or two_res):
            return None
        Lc = self._embedding_box(L, two_res)
        if Lc is None:
            return None
        dev = self.device

        embed = self._embed
        has_clash = rec_forb is not None
        rec_e = [embed(rec[0], Lc)] + ([embed(rec[1], Lc // 2)] if two_res else [])
        rf_e = embed(torch.as_tensor(rec_forb, dtype=torch.float32).reshape(L, L, L), Lc) if has_clash else None
        eng = self._make_engine(rec_e, rf_e, batch_size, params, inner_box=L)
        if eng is None:
            return None
        self.engine_box = Lc
        lf_e = None
        if has_clash:
            lf_e = embed(torch.as_tensor(lig_forb, dtype=torch.float32).reshape(L, L, L), Lc) if clash_provider is None \
                else torch.zeros(Lc, Lc, Lc)
        eng.set_ligand(embed(lig[0], Lc), lf_e, embed(lig[1], Lc // 2) if two_res else None)
        forb_e = torch.zeros(batch_size, Lc, Lc, Lc, dtype=torch.float32, device=dev) if clash_provider is not None else None

        def provider(Rb):          # the re-projected clash volumes of the batch (Docker.py:221-224), into the corner of Lc^3
            n = Rb.shape[0]
            forb_e[:n, :L, :L, :L] = clash_provider(Rb).reshape(n, L, L, L)
            return forb_e[:n]

        eng.clash_provider = provider if clash_provider is not None else None
        eng.reset_top()
        eng.search(self.rot.R[ids], rot_ids=ids)          # the engine's top-K works on the gathered (2L)^3 grid (engine.window)
        self.engine = eng
        entries = eng.top_entries()
        eng.clash_provider = None
        return entries

    @staticmethod
    def _ops_path_ok(rec, params):
        """The stand-alone ops + the HIP per-voxel filter kernel cover an MLP filter of hidden width
        <= 64 over one or two resolutions (dlpd_filter_mask)."""
        return params is not None and params[0].shape[0] <= 64 and len(rec) <= 2

    def _batch_inputs(self, rec, lig, rec_forb, lig_forb, clash_provider):
        dev = self.device
        L = rec[0].shape[-1]
        rec_d = [v.to(dev) for v in rec]
        lig_d = [v.to(dev) for v in lig]
        rf = lf = None
        if rec_forb is not None:
            rf = torch.as_tensor(rec_forb, dtype=torch.float32).reshape(1, 1, L, L, L).to(dev)
            if clash_provider is None:
                lf = torch.as_tensor(lig_forb, dtype=torch.float32).reshape(1, 1, L, L, L).to(dev)
        return rec_d, lig_d, rf, lf

    def _dock_volumes_multires(self, rec, lig, rec_forb, lig_forb, batch_size, ids, clash_provider=None, params=None):
        """Reference-shaped loop on the stand-alone ops: rotate, clash correlation, per-resolution
        correlation, HIP filter kernel (upsample + concat + MLP + mask multiply), device top-K."""
        from deeplocalproteindocking_amd.ops import VolumeConvolution, VolumeRotation, filter_volumes
        dev = self.device
        model = self.docking_model
        emb = self.embed_uncompiled_boxes
        rotate, conv_noclip = _PivotRotation(self), VolumeConvolution(lib=self._lib, embed=emb)
        convolve = getattr(model, "convolve", None) or VolumeConvolution(clip=getattr(model, "clip", 5.0), lib=self._lib)
        if isinstance(convolve, VolumeConvolution):        # the reference's op: same clip, this Docker's box policy
            convolve = VolumeConvolution(clip=convolve.clip, lib=self._lib, embed=emb, clip_mode=self.conventions.clip_mode)
        rec_d, lig_d, rf, lf = self._batch_inputs(rec, lig, rec_forb, lig_forb, clash_provider)
        L = rec[0].shape[-1]
        top = DeviceTopList(self.max_conf, batch_size, dev, self._library())
        top.reset()
        has_clash = rec_forb is not None
        R_all = self.rot.R
        W1, b1, W2, b2 = params if params is not None else fused_filter_parameters(model)
        for beg in range(0, len(ids), batch_size):
            bid = ids[beg:beg + batch_size]
            nb = len(bid)
            Rb = R_all[bid].to(device=dev, dtype=torch.float32).contiguous()
            lig_rot = [rotate(v.unsqueeze(0).expand(nb, -1, -1, -1, -1).contiguous(), Rb) for v in lig_d]
            rec_b = [v.unsqueeze(0).expand(nb, -1, -1, -1, -1).contiguous() for v in rec_d]
            convolved = [convolve(r, l) for r, l in zip(rec_b, lig_rot)]
            norm = None
            if has_clash:
                lfr = (clash_provider(Rb).reshape(nb, 1, L, L, L).contiguous() if clash_provider is not None
                       else rotate(lf.expand(nb, -1, -1, -1, -1).contiguous(), Rb))
                norm = conv_noclip(rf.expand(nb, -1, -1, -1, -1).contiguous(), lfr).squeeze(1).contiguous()
            V = filter_volumes(convolved, W1, b1, W2, float(b2.reshape(-1)[0]), mask_norm=norm,
                               threshold=model.threshold_clash, lib=self._lib)
            top.select(V.reshape(nb, -1), nb)
            top.merge(torch.as_tensor(bid, dtype=torch.int32).to(dev), nb)
        return top.entries()

    def _dock_volumes_generic(self, rec, lig, rec_forb, lig_forb, batch_size, ids, clash_provider=None):
        """The loop body of Docker.py:218-232 with the scoring model really CALLED --
        ``V = self.docking_model(receptor_volumes, ligand_volumes_rotated)`` (Docker.py:229), whatever
        module that is -- between the stand-alone HIP ops (volume rotation, clash correlation) and the
        device top-K.  This is the path of a user-defined filter / docking model."""
        from deeplocalproteindocking_amd.ops import VolumeConvolution, VolumeRotation
        dev = self.device
        model = self.docking_model
        rotate, conv_noclip = _PivotRotation(self), VolumeConvolution(lib=self._lib, embed=self.embed_uncompiled_boxes)
        rec_d, lig_d, rf, lf = self._batch_inputs(rec, lig, rec_forb, lig_forb, clash_provider)
        L = rec[0].shape[-1]
        top = DeviceTopList(self.max_conf, batch_size, dev, self._library())
        top.reset()
        has_clash = rec_forb is not None
        R_all = self.rot.R
        with torch.no_grad():
            for beg in range(0, len(ids), batch_size):
                bid = ids[beg:beg + batch_size]
                nb = len(bid)
                Rb = R_all[bid].to(device=dev, dtype=torch.float32).contiguous()
                lig_rot = [rotate(v.unsqueeze(0).expand(nb, -1, -1, -1, -1).contiguous(), Rb) for v in lig_d]
                rec_b = [v.unsqueeze(0).expand(nb, -1, -1, -1, -1).contiguous() for v in rec_d]
                V = model(rec_b, lig_rot).reshape(nb, 2 * L, 2 * L, 2 * L).to(torch.float32)
                if has_clash:
                    lfr = (clash_provider(Rb).reshape(nb, 1, L, L, L).contiguous() if clash_provider is not None
                           else rotate(lf.expand(nb, -1, -1, -1, -1).contiguous(), Rb))
                    norm = conv_noclip(rf.expand(nb, -1, -1, -1, -1).contiguous(), lfr).squeeze(1)
                    V = torch.lt(norm, model.threshold_clash).to(dtype=torch.float32) * V
                V = V.contiguous()
                top.select(V.reshape(nb, -1), nb)
                top.merge(torch.as_tensor(bid, dtype=torch.int32).to(dev), nb)
        return top.entries()

    def _gather(self, entries):
        """One all-gather of the fixed-size per-rank lists + deterministic merge (every rank)."""
        return all_gather_top_entries(entries, self.max_conf, self.world_size, self.process_group, self.device,
                                      always=self.collectives_with_one_rank)

    # ------------------------------------------------------------------ PDB-level entries
    def _need_backend(self):
        """The atom front end (TorchProteinLibrary's PDB2CoordsUnordered / Coords2TypedCoords /
        CoordsTranslate / CoordsRotate / TypedCoords2Volume, Docker.py:29-39): this build's
        ``CoordsBackend`` unless the constructor was given another one."""
        if self.coords_backend is None:
            from deeplocalproteindocking_amd.Utils.FullAtom import CoordsBackend
            self.coords_backend = CoordsBackend(lib=self._lib, splat=self.conventions.splat,
                                                atom_types=self.conventions.atom_types)
        return self.coords_backend

    def load_batch(self, filenames, bbox_center=True):
        """Docker.py:49-61 through the coords backend."""
        be = self._need_backend()
        coords, chains, resnames, resnums, atomnames, num_atoms = be.pdb2coords(filenames)
        coords, num_atoms_of_type, offsets = be.assign_types(coords, resnames, atomnames, num_atoms)
        num_atoms = getattr(be, "last_num_typed", num_atoms)        # untyped atoms (hydrogens) were dropped
        a, b = be.get_bbox(coords, num_atoms)
        if bbox_center:
            translation = -(a + b) * 0.5 + self.box_length / 2.0
        else:
            translation = -(a + b) * 0.5
        coords = be.translate(coords, translation, num_atoms)
        return coords, num_atoms_of_type, offsets, translation, num_atoms

    def prepare(self, ureceptor, uligand, group="SE3", slot=0, stream=None):
        """The rotation-independent part of ``dockSE3`` / ``dockE3`` for one target (Docker.py:184-209 / :135-161) ->
        ``PreparedPair``.  ``slot`` names the engine it fills (0 / 1), so that a sweep over targets (local_test.py:57-75)
        can prepare target n + 1 -- from a host thread of its own -- while target n is being searched in the other engine;
        its device work goes to the CALLER'S current stream, i.e. between two batches of the running search, never beside
        them.  No collective is issued here (the random receptor rotation is shared by ``dock*`` / the constructor, on the
        caller's thread).
        ``stream`` is a DIAGNOSTIC argument (scripts/prepare_race_probe.py) and warns: a preparation on a second stream puts the
        plugin's matrix-instruction convolution on the same CUs as the search's kernels, which on this hardware changes low
        mantissa bits of a few scores (inside the 1e-4 parity band, but the .dat is then not byte-reproducible; an OPEN
        defect, not root-caused: DESIGN.md section 8, EXPERIMENTS.md R5).  The product never passes it."""
        import contextlib
        import time
        import warnings
        if stream is not None:
            warnings.warn("dlpd: Docker.prepare(stream=...) runs the representation beside the search's kernels; on this hardware "
                          "that co-residency perturbs low bits of a few scores (results stay inside the 1e-4 parity band but are "
                          "not byte-reproducible) -- diagnostic use only", RuntimeWarning, stacklevel=2)
        t0 = time.perf_counter()
        if group not in ("SE3", "E3"):
            raise Exception("Unknown equivariance group", group)
        be = self._need_backend()
        model = self.docking_model
        model.eval()
        dev, L, res = self.device, self.box_size, self.resolution
        p = PreparedPair(group, ureceptor, uligand, slot)
        on_stream = stream is not None and dev.type == "cuda"
        with (torch.cuda.stream(stream) if on_stream else contextlib.nullcontext()), torch.no_grad():
            rcoords, rnat, roff, rT, rnatoms = self.load_batch([ureceptor], bbox_center=False)
            lcoords, lnat, loff, lT, lnatoms = self.load_batch([uligand], bbox_center=False)
            if self.randomize_rot:
                rcoords = be.rotate(rcoords, self.randR, rnatoms)
            rcoords = be.translate(rcoords, self.box_center, rnatoms)
            p.receptor = be.project(rcoords, rnat, roff, L, res, dev)
            p.receptor_volumes = model.representation(p.receptor)
            p.receptor_forbidden = p.receptor.sum(dim=1)[0]
            p.ligand_atoms = be.to_device(lcoords, lnat, loff, dev)             # origin-centred ligand atoms
            if group == "SE3":
                lcoords_trans = be.translate(lcoords, self.box_center, lnatoms)
                ligand = be.project(lcoords_trans, lnat, loff, L, res, dev)
                p.ligand_volumes = model.representation(ligand)
            # the fused engine of a compiled box: receptor spectrum (+ ligand layouts) now, not at the start of the search
            rec = [v.reshape((-1,) + tuple(v.shape[-3:])) for v in p.receptor_volumes]
            params = fused_filter_parameters(model)
            if params is not None and rec[0].shape[-1] == L:
                eng = self._make_engine(rec, p.receptor_forbidden, self.launch_batch, params, slot=slot)
                if eng is not None and group == "SE3":
                    lig = [v.reshape((-1,) + tuple(v.shape[-3:])) for v in p.ligand_volumes]
                    # (the clash channel comes from re-projected atoms: the stored forbidden volume is never read)
                    eng.set_ligand(lig[0], torch.zeros(L, L, L), lig[1] if eng.C1 > 0 else None)
                p.engine = eng
                if eng is not None:
                    eng.holds_pair = p                   # (checked when the pair is docked: a later prepare() into the same
                                                         #  slot replaces the receptor / ligand this engine holds)
            if on_stream:
                p.ready = torch.cuda.Event()
                p.ready.record(stream)
        p.seconds = time.perf_counter() - t0
        return p

    def _prepared_for(self, prepared, ureceptor, uligand, group):
        if prepared is None:
            return self.prepare(ureceptor, uligand, group)
        if (prepared.group, prepared.ureceptor, prepared.uligand) != (group, ureceptor, uligand):
            raise Exception("Prepared pair does not belong to this call", prepared.group, prepared.ureceptor, prepared.uligand)
        if prepared.engine is not None and getattr(prepared.engine, "holds_pair", None) is not prepared:
            raise Exception("Prepared pair was overwritten: another pair was prepared into engine slot %d after it" % prepared.slot)
        prepared.wait(self.device)
        return prepared

    def dockSE3(self, ureceptor, uligand, batch_size, prepared=None):
        """Docker.py:184-238: representations computed once, ligand volumes rotated on the GPU, and
        the ligand forbidden volume re-projected from the rotated ATOMS every batch (Docker.py:221-224)
        -- by one kernel that rotates on the fly, without the reference's per-batch host round trip.
        ``batch_size`` (2 in local_test.py:71) only sizes the calls of a model that has to be called
        (generic path); the fused pipeline always launches ``self.launch_batch`` rotations.
        prepared: the ``PreparedPair`` of this target if ``prepare`` already ran (a sweep preparing targets ahead)."""
        be = self._need_backend()
        self.top_list = []
        self.docking_model.eval()
        self._share_random_rotation()
        p = self._prepared_for(prepared, ureceptor, uligand, "SE3")
        L, res, dev = self.box_size, self.resolution, self.device
        with torch.no_grad():
            lc, ln, lo = p.ligand_atoms

            def provider(Rb):   # rotate about the origin, translate to the box centre, project, sum types
                return be.project(lc, ln, lo, L, res, dev, R=Rb, shift=self.box_center, sum_types=True)

            self.dock_volumes(p.receptor_volumes, p.ligand_volumes, p.receptor_forbidden, None, batch_size=None,
                              clash_provider=provider, model_batch=batch_size, prepared=p)


    def dockE3(self, ureceptor, uligand, batch_size, prepared=None):
        """Docker.py:135-182: the ligand is rotated in coordinate space and re-projected and
        re-represented every batch (the plugin's cost), then scored by the same kernels: the fused
        engine takes the batch's volumes as they are (no volume rotation); the stand-alone ops, or a
        call of the model itself, where the engine has no layout for the model (see dock_volumes).
        The two halves of a batch run one after the other on the caller's stream (``_dockE3_fused``)."""
        be = self._need_backend()
        from deeplocalproteindocking_amd.ops import VolumeConvolution, filter_volumes
        self.top_list = []
        model = self.docking_model
        model.eval()
        dev = self.device
        self._share_random_rotation()
        p = self._prepared_for(prepared, ureceptor, uligand, "E3")
        ids = self.shard(self.rot.R.shape[0])
        L = self.box_size
        with torch.no_grad():
            receptor, receptor_volumes = p.receptor, p.receptor_volumes
            lc, ln, lo = p.ligand_atoms
            rec = [v.reshape((-1,) + tuple(v.shape[-3:])) for v in receptor_volumes]
            params = fused_filter_parameters(model)
            eng = p.engine
            Lc = None
            if eng is None and params is not None and self.embed_uncompiled_boxes and \
                    not self._library().call("dlpd_grid_supported", int(L)):
                # a box without a compiled plan: the batch's volumes in the corner of the next compiled box, scored by
                # the same engine; its top-K stage gathers the reference's grid (see _dock_volumes_embedded)
                two_res = len(rec) == 2 and rec[1].shape[-1] * 2 == L
                Lc = self._embedding_box(L, two_res) if (len(rec) == 1 or two_res) else None
                if Lc is not None:
                    eng = self._make_engine([self._embed(rec[0], Lc)] + ([self._embed(rec[1], Lc // 2)] if two_res else []),
                                            self._embed(receptor.sum(dim=1)[0], Lc), self.launch_batch, params, inner_box=L)
                    Lc = Lc if eng is not None else None
                    self.engine_box = Lc
            self.path = ("embedded" if Lc else "fused") if eng is not None else ("ops" if self._ops_path_ok(rec, params) else "call")
            nbatch = self.launch_batch if self.path != "call" else int(batch_size)
            if eng is not None:
                eng.reset_top()
                self.engine = eng
            else:
                emb = self.embed_uncompiled_boxes
                conv_noclip = VolumeConvolution(lib=self._lib, embed=emb)
                convolve = getattr(model, "convolve", None) or VolumeConvolution(clip=getattr(model, "clip", 5.0),
                                                                                 lib=self._lib)
                if isinstance(convolve, VolumeConvolution):
                    convolve = VolumeConvolution(clip=convolve.clip, lib=self._lib, embed=emb,
                                                 clip_mode=self.conventions.clip_mode)
                top = DeviceTopList(self.max_conf, nbatch, dev, self._library())
                top.reset()
                receptor_forbidden = receptor.sum(dim=1).unsqueeze(dim=1).contiguous()

            def represent(bid):
                """rotate + translate + project in one kernel (Docker.py:163-165), then the plugin (Docker.py:166-167)"""
                Rb = self.rot.R[bid].to(device=dev, dtype=torch.float32).contiguous()
                # (_project_cells: set by _dockE3_fused when the plugin and the engine go by occupancy maps -- the projection
                #  then writes only the cells the atoms reach)
                ligand = be.project(lc, ln, lo, L, self.resolution, dev, R=Rb, shift=self.box_center,
                                    cells=bool(getattr(self, "_project_cells", False)))
                # the forbidden volume = the sum over the atom types (Docker.py:169): projected once more with the types summed in
                # the accumulator (33 MB; exact) instead of a pass over the eleven volumes
                ligand.dlpd_type_sum = be.project(lc, ln, lo, L, self.resolution, dev, R=Rb, shift=self.box_center, sum_types=True)[:, 0]
                return ligand, model.representation(ligand)

            batches = [ids[beg:beg + nbatch] for beg in range(0, len(ids), nbatch)]
            if eng is not None:
                self._dockE3_fused(eng, batches, represent, Lc, nbatch)
                batches = []
            for bid in batches:
                nb = len(bid)
                ligand, ligand_volumes = represent(bid)
                bid_dev = torch.as_tensor(bid, dtype=torch.int32).to(dev)
                ligand_forbidden = ligand.sum(dim=1).unsqueeze(dim=1).contiguous()
                norm = conv_noclip(receptor_forbidden.expand(nb, -1, -1, -1, -1).contiguous(), ligand_forbidden)
                norm = norm.squeeze(1).contiguous()
                rec_b = [v.expand(nb, -1, -1, -1, -1).contiguous() for v in receptor_volumes]
                if self.path == "ops":
                    W1, b1, W2, b2 = params
                    convolved = [convolve(r, l) for r, l in zip(rec_b, ligand_volumes)]
                    V = filter_volumes(convolved, W1, b1, W2, float(b2.reshape(-1)[0]), mask_norm=norm,
                                       threshold=model.threshold_clash, lib=self._lib)
                else:                                              # Docker.py:171-175 with the model called
                    V = model(rec_b, ligand_volumes).reshape(nb, 2 * L, 2 * L, 2 * L).to(torch.float32)
                    V = (torch.lt(norm, model.threshold_clash).to(dtype=torch.float32) * V).contiguous()
                top.select(V.reshape(nb, -1), nb)
                top.merge(bid_dev, nb)
        entries = self._gather(eng.top_entries() if eng is not None else top.entries())
        self.top_list = DeviceTopList.to_top_list(entries, 2 * self.box_size)
        self.write_conformations()

    def _dockE3_fused(self, eng, batches, represent, Lc, nbatch):
        """The batch loop of dockE3 on the fused engine: projection + representation of a batch, then the engine's half,
        on ONE stream.  (Round 5 built the plugin's half of batch i + 1 on a second stream beside the engine's half of
        batch i: no gain -- 15.08 against 15.03 ms per launch, both halves fill the chip -- and that co-residency of the
        plugin's matrix-instruction convolution with the pipeline's LDS kernels is exactly what perturbs the pipeline's low
        bits on this hardware, EXPERIMENTS.md R5; removed.)"""
        dev = self.device
        ebuf = {}

        def volumes_of(ligand, ligand_volumes, nb, slot):
            forb = getattr(ligand, "dlpd_type_sum", None)
            vols = (ligand_volumes[0], forb if forb is not None else ligand.sum(dim=1), ligand_volumes[1] if eng.C1 else None)
            if Lc:
                # the batch's volumes into the corner of buffers that were zeroed ONCE (not three allocations and zero
                # fills per batch); two sets when the plugin runs ahead of the engine
                if slot not in ebuf:
                    ebuf[slot] = [torch.zeros((nbatch,) + tuple(v.shape[1:-3]) + (le, le, le), dtype=torch.float32, device=dev)
                                  if v is not None else None for v, le in zip(vols, (Lc, Lc, Lc // 2))]
                for buf, v in zip(ebuf[slot], vols):
                    if v is not None:
                        buf[:nb][..., :v.shape[-3], :v.shape[-2], :v.shape[-1]] = v
                vols = tuple(None if buf is None else buf[:nb] for buf in ebuf[slot])
            return vols

        # A plugin whose layers carry occupancy maps (E3MultiResRepr4x4: bias-free convolutions) hands the maps on with its
        # volumes; the engine's K1 then never reads an empty cell, so the plugin need not WRITE the tiles it skips
        # (``outputs_with_maps``).  Not with an embedded box: the copy into the larger box reads every voxel.
        import contextlib
        rep = getattr(self.docking_model, "representation", None)
        # (clip_mode "input" clamps every voxel of the batch's volumes: maps are not combined with it)
        maps_ok = (not Lc) and eng._in_clip is None
        with_maps = maps_ok and bool(getattr(rep, "supports_unwritten_outputs", False)) and self.unwritten_activations
        from deeplocalproteindocking_amd import ops as _ops
        cells_ok = with_maps and bool(getattr(rep, "use_tile_occupancy", False)) and bool(getattr(rep, "use_hip_conv", False)) and \
            _ops.CONV_PRECISION == "split_bf16" and self.device.type == "cuda"
        for bid in batches:
            self._project_cells = cells_ok
            try:
                with (rep.outputs_with_maps() if with_maps else contextlib.nullcontext()):
                    ligand, ligand_volumes = represent(bid)
            finally:
                self._project_cells = False
            bid_dev = torch.as_tensor(bid, dtype=torch.int32).to(dev)
            occ = None
            if maps_ok:
                o0 = getattr(ligand_volumes[0], "dlpd_occupancy", None)
                o1 = getattr(ligand_volumes[1], "dlpd_occupancy", None) if eng.C1 else None
                occ = (o0, o1) if (o0 is not None or o1 is not None) else None
            eng.step(None, bid_dev, volumes=volumes_of(ligand, ligand_volumes, len(bid), 0), occupancy=occ)
