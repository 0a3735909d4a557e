"""Transport of the sharded path: the two callbacks of include/hymls_mi.h's hymls_mi_comm on top of
torch.distributed (backend "nccl" = RCCL over xGMI for device buffers; "gloo" for the host-side setup
exchanges and for the CPU tests).  One process per GPU.

The library decides what goes where (halo of the interior layer and of the separators around every
SpMV, Schur records at setup, V-sum hand-off per level); this file only moves bytes:
  alloc(bytes)          -> a torch tensor the transport can address (exchange arenas)
  alltoallv(send, ...)  -> dist.all_to_all_single on views of those tensors, ordered on the handle's stream
"""
import ctypes as C

import numpy as np
import torch
import torch.distributed as dist

from .api import A2A_FN, ALLOC_FN, _Comm


def rank_grid(world):
    """boxes of the grid per direction for `world` ranks: 2 -> (2,1,1), 4 -> (2,2,1), 8 -> (2,2,2), 6 -> (3,2,1), 12 -> (3,2,2) ...
    The prime factors of `world`, largest first, go to the direction with the fewest boxes so far (powers of two: the repeated
    halving x, y, z of the reference's CreatePIDMap, src/HYMLS_BasePartitioner.cpp:361-586).  The library checks that the
    subdomains of every level divide evenly over the boxes (hymls_mi_set_comm)."""
    assert world >= 1
    factors, w, p = [], world, 2
    while w > 1:
        while w % p == 0:
            factors.append(p)
            w //= p
        p += 1
    g = [1, 1, 1]
    for f in sorted(factors, reverse=True):
        d = min(range(3), key=lambda k: (g[k], k))
        g[d] *= f
    return tuple(g)


def transport_selftest(device, backend):
    """Collective: an uneven all_to_all_single of float64 device buffers on a side stream, the operation the
    sharded ApplyInverse relies on.  Returns None if it works on every rank, else a short reason."""
    device = torch.device(device)
    rank, world = dist.get_rank(), dist.get_world_size()
    err = None
    try:
        sc = [(rank + q) % 3 for q in range(world)]            # what I send to q
        rc = [(q + rank) % 3 for q in range(world)]            # what q sends to me
        inp = torch.cat([torch.full((c,), float(rank * 100 + q), dtype=torch.float64, device=device) for q, c in enumerate(sc)]
                        + [torch.empty(0, dtype=torch.float64, device=device)])
        out = torch.full((sum(rc),), -1.0, dtype=torch.float64, device=device)
        if device.type == "cuda" and backend != "gloo":
            st = torch.cuda.Stream(device=device)
            ext = torch.cuda.ExternalStream(st.cuda_stream, device=device)
            with torch.cuda.stream(ext):
                dist.all_to_all_single(out, inp, rc, sc)
            st.synchronize()
        elif device.type == "cuda":
            co = torch.empty(out.numel(), dtype=torch.float64)
            dist.all_to_all_single(co, inp.cpu(), rc, sc)
            out.copy_(co)
        else:
            dist.all_to_all_single(out, inp, rc, sc)
        exp = torch.cat([torch.full((c,), float(q * 100 + rank), dtype=torch.float64) for q, c in enumerate(rc)]
                        + [torch.empty(0, dtype=torch.float64)])
        if not torch.equal(out.cpu(), exp):
            err = "wrong data"
    except Exception as e:  # pragma: no cover
        err = type(e).__name__ + ": " + str(e)[:120]
    flag = torch.tensor([1.0 if err else 0.0], dtype=torch.float64, device=device if backend != "gloo" else "cpu")
    try:
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)
    except Exception as e:  # pragma: no cover
        return err or (type(e).__name__ + ": " + str(e)[:120])
    if flag.item() > 0:
        return err or "failed on another rank"
    return None


class RcclComm:
    """The library's built-in transport (include/hymls_mi.h: hymls_mi_set_comm_rccl; hymls_amd/csrc/comm_rccl.cpp): RCCL
    send/recv groups on the handle's stream, no Python in ApplyInverse.  This class only bootstraps the ncclComm_t:
    rank 0 draws the unique id, torch.distributed (any backend, host tensors over gloo or device tensors over nccl)
    broadcasts its 128 bytes, every rank calls ncclCommInitRank through the library.  Without torch.distributed
    (rank = 0, size = 1) the communicator has a single rank."""

    def __init__(self, device_index=0, lib=None, group=None):
        from .api import load_library
        self._lib = lib or load_library()
        use_dist = dist.is_available() and dist.is_initialized()
        self.rank = dist.get_rank(group) if use_dist else 0
        self.size = dist.get_world_size(group) if use_dist else 1
        idbuf = C.create_string_buffer(128)
        if self.rank == 0:
            ierr = self._lib.hymls_mi_rccl_unique_id(idbuf)
            if ierr:
                raise RuntimeError("hymls_mi_rccl_unique_id failed (%d)" % ierr)
        if use_dist and self.size > 1:
            on_dev = dist.get_backend(group) == "nccl"
            t = torch.tensor(list(idbuf.raw), dtype=torch.uint8, device=torch.device("cuda", device_index) if on_dev else "cpu")
            dist.broadcast(t, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
            idbuf = C.create_string_buffer(bytes(t.cpu().tolist()), 128)
        comm = C.c_void_p()
        ierr = self._lib.hymls_mi_rccl_comm_init(idbuf, self.rank, self.size, device_index, C.byref(comm))
        if ierr:
            raise RuntimeError("hymls_mi_rccl_comm_init failed (%d)" % ierr)
        self.nccl_comm = comm
        self.error = None

    def attach(self, prec):
        import weakref
        self._prec = weakref.proxy(prec)      # (no reference cycle: `del P` has to free the handle's device memory at once)

    def close(self):
        if self.nccl_comm is not None:
            self._lib.hymls_mi_rccl_comm_destroy(self.nccl_comm)
            self.nccl_comm = None


class TorchComm:
    """device: torch.device of this rank's buffers ('cpu' with the test-only host simulator).
    device_group: process group for device buffers (default group); host_group: gloo group for
    host buffers (created here when the default backend cannot move host memory)."""

    def __init__(self, device, device_group=None, host_group=None):
        assert dist.is_initialized()
        self.device = torch.device(device)
        self.rank = dist.get_rank()
        self.size = dist.get_world_size()
        self.device_group = device_group
        backend = dist.get_backend(device_group)
        if host_group is None and backend != "gloo":
            host_group = dist.new_group(backend="gloo")   # collective: every rank constructs its TorchComm
        self.host_group = host_group if backend != "gloo" else (host_group or device_group)
        import os
        self.host_chunk = int(os.environ.get("HYMLS_MI_HOST_CHUNK_BYTES", str(256 << 20)))   # bytes per peer and round
        self._arenas = []          # (ptr, nbytes, tensor)
        self._stream = None
        self._a2a = A2A_FN(self._alltoallv)
        self._alloc = ALLOC_FN(self._alloc_arena)
        self.c_struct = _Comm(None, self.rank, self.size, self._a2a, self._alloc)
        self.error = None

    def attach(self, prec):
        import weakref
        self._prec = weakref.proxy(prec)

    # ---- callbacks (no exception may cross the C boundary)
    def _alloc_arena(self, ctx, nbytes):
        try:
            t = torch.empty((nbytes + 7) // 8, dtype=torch.float64, device=self.device)
            self._arenas.append((t.data_ptr(), t.numel() * 8, t))
            return t.data_ptr()
        except Exception as e:  # pragma: no cover
            self.error = e
            return None

    def _view(self, ptr, nbytes):
        for base, size, t in self._arenas:
            if base <= ptr and ptr + nbytes <= base + size:
                off = (ptr - base) // 8
                return t[off:off + nbytes // 8]
        raise RuntimeError("exchange buffer outside the arenas")

    def _alltoallv(self, ctx, send, scnt, recv, rcnt, elem_bytes, on_device):
        try:
            sc = [int(scnt[i]) for i in range(self.size)]
            rc = [int(rcnt[i]) for i in range(self.size)]
            if on_device:
                assert elem_bytes == 8
                inp = self._view(send or 0, sum(sc) * 8) if sum(sc) else torch.empty(0, dtype=torch.float64, device=self.device)
                out = self._view(recv or 0, sum(rc) * 8) if sum(rc) else torch.empty(0, dtype=torch.float64, device=self.device)
                if self.device.type == "cuda" and dist.get_backend(self.device_group) == "gloo":
                    # several ranks sharing one GPU (tests): RCCL cannot do that, stage through the host
                    torch.cuda.synchronize()
                    ci, co = inp.cpu(), torch.empty(out.numel(), dtype=torch.float64)
                    dist.all_to_all_single(co, ci, rc, sc, group=self.device_group)
                    out.copy_(co)
                    torch.cuda.synchronize()
                elif self.device.type == "cuda":
                    if self._stream is None:
                        self._stream = torch.cuda.ExternalStream(self._prec.stream(), device=self.device)
                    with torch.cuda.stream(self._stream):
                        dist.all_to_all_single(out, inp, rc, sc, group=self.device_group)
                else:
                    dist.all_to_all_single(out, inp, rc, sc, group=self.device_group)
            else:
                sb, rb = [c * elem_bytes for c in sc], [c * elem_bytes for c in rc]
                nb_s, nb_r = sum(sb), sum(rb)
                inp = torch.frombuffer((C.c_char * max(nb_s, 1)).from_address(send), dtype=torch.uint8)[:nb_s]
                out = torch.frombuffer((C.c_char * max(nb_r, 1)).from_address(recv), dtype=torch.uint8)[:nb_r]
                # large setup exchanges (the reduced matrix of a 256^3 run is GBs) go in rounds of bounded size; the
                # number of rounds has to be the same on every rank
                big = torch.tensor([max(sb + rb + [0])], dtype=torch.int64)
                dist.all_reduce(big, op=dist.ReduceOp.MAX, group=self.host_group)
                rounds = max(1, -(-int(big.item()) // self.host_chunk))
                if rounds == 1:
                    dist.all_to_all_single(out, inp, rb, sb, group=self.host_group)
                else:
                    so = np.concatenate([[0], np.cumsum(sb)]).astype(np.int64)
                    ro = np.concatenate([[0], np.cumsum(rb)]).astype(np.int64)
                    ch = self.host_chunk
                    for r in range(rounds):
                        ss = [min(max(b - r * ch, 0), ch) for b in sb]
                        rs = [min(max(b - r * ch, 0), ch) for b in rb]
                        ti = torch.cat([inp[so[q] + r * ch: so[q] + r * ch + ss[q]] for q in range(self.size)])
                        to = torch.empty(sum(rs), dtype=torch.uint8)
                        dist.all_to_all_single(to, ti, rs, ss, group=self.host_group)
                        o = 0
                        for q in range(self.size):
                            out[ro[q] + r * ch: ro[q] + r * ch + rs[q]] = to[o:o + rs[q]]
                            o += rs[q]
            return 0
        except Exception as e:
            self.error = e
            return -1
