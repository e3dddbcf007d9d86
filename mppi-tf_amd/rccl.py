"""RCCL reached through its C entry points, for the ONE-call sharded step (mppi_shard_step, include/mppi_c.h).

mppi_shard_step takes the caller's collectives as two function pointers with exactly ncclAllGather's / ncclAllReduce's signatures plus
the communicator, and calls them on the step's stream between its own kernels: the whole sharded step is then one C call, instead of
three ctypes calls and a torch.distributed call from Python (29.9 us per step with one rank, very likely host-bound).

The library here is the librccl.so the process ALREADY holds (torch's: backend "nccl" is RCCL on ROCm) — never a second copy. The
communicator is the controller's own (ncclCommInitRank; the unique id travels over the job's torch.distributed group), so nothing of
torch's ProcessGroupNCCL state is shared: its collectives run on its own internal stream, these on the caller's.
This is plumbing (a communicator and two addresses); no compute happens here.
"""
import ctypes as C
import os

from ._lib import Collectives

_rccl = None


class UniqueId(C.Structure):
    _fields_ = [("internal", C.c_char * 128)]  # ncclUniqueId: NCCL_UNIQUE_ID_BYTES = 128


def _mapped_rccl_path():
    """the librccl the process has mapped (after `import torch`), or None"""
    try:
        with open("/proc/self/maps") as f:
            for line in f:
                p = line.split()[-1]
                if "librccl" in os.path.basename(p):
                    return p
    except OSError:
        pass
    return None


def load():
    """-> the ctypes library of RCCL (raises OSError when there is none)"""
    global _rccl
    if _rccl is not None:
        return _rccl
    import torch
    cands = [_mapped_rccl_path(), os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so"), "librccl.so", "librccl.so.1"]
    err = None
    for p in cands:
        if not p:
            continue
        try:
            lib = C.CDLL(p)
        except OSError as e:
            err = e
            continue
        lib.ncclGetUniqueId.restype, lib.ncclGetUniqueId.argtypes = C.c_int, [C.POINTER(UniqueId)]
        lib.ncclCommInitRank.restype, lib.ncclCommInitRank.argtypes = C.c_int, [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
        lib.ncclCommDestroy.restype, lib.ncclCommDestroy.argtypes = C.c_int, [C.c_void_p]
        lib.ncclGetErrorString.restype, lib.ncclGetErrorString.argtypes = C.c_char_p, [C.c_int]
        _rccl = lib
        return lib
    raise OSError("no librccl.so in this process or on the loader path: %s" % err)


class RcclComm:
    """One rank's own communicator + the mppi_collectives struct that points at ncclAllGather / ncclAllReduce.

    Collective over `group` (every rank constructs it at the same point): rank 0 draws the unique id, the job's group carries it."""

    def __init__(self, rank, world, group=None):
        import torch.distributed as dist
        lib = self.lib = load()
        uid = UniqueId()
        if rank == 0:
            self._check(lib.ncclGetUniqueId(C.byref(uid)), "ncclGetUniqueId")
        if world > 1:
            box = [C.string_at(C.byref(uid), 128) if rank == 0 else None]
            src = dist.get_global_rank(group, 0) if group is not None else 0
            dist.broadcast_object_list(box, src=src, group=group)
            C.memmove(C.byref(uid), box[0], 128)
        self.comm = C.c_void_p()
        self._check(lib.ncclCommInitRank(C.byref(self.comm), world, uid, rank), "ncclCommInitRank")
        addr = lambda fn: C.cast(fn, C.c_void_p).value
        self.coll = Collectives(all_gather=addr(lib.ncclAllGather), all_reduce=addr(lib.ncclAllReduce), comm=self.comm)

    def _check(self, rc, what):
        if rc != 0:
            raise RuntimeError("%s: %s" % (what, (self.lib.ncclGetErrorString(rc) or b"?").decode()))

    def close(self):
        comm, self.comm = getattr(self, "comm", None), None
        if comm:
            self.lib.ncclCommDestroy(comm)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
