"""CTU-row bands: which rows of a picture a rank owns and what it exchanges with its neighbours.

One picture is sharded over the GPUs of a node by contiguous CTU rows (the reference runs one job
per CTU row with a dependency on the row above, src/encoderstate.c:1085-1189).  Every rank keeps FULL-SIZE planes,
produces only the rows it owns and receives the halo rows into the same picture coordinates, so every kernel keeps
its picture-border behaviour and a band of a picture is bit-identical to the same rows of the whole-frame run.

The geometry comes from the C ABI (uvghip_band_plan, host only); this module turns it into exchange lists
    [(peer, [(tensor, row0, row1) to send ...], [(tensor, row0, row1) to receive ...]), ...]
over whatever 2-D row-major tensors it is given (HIP tensors in the product, CPU tensors in the gloo test), and
offers three transports for them:
    RcclTransport   uvghip_comm_exchange / _allreduce_i64: ncclGroupStart + ncclSend/ncclRecv + ncclGroupEnd on the
                    caller's HIP stream (the product path)
    TorchTransport  torch.distributed.batch_isend_irecv (gloo on CPU: the 2-process schedule test)
    emulate()       several emulated ranks inside one process: plain row copies between their tensors (the 1-GPU
                    bit-exactness test of the banded filters against the whole-frame kernels)
"""
import ctypes

from . import lib as _lib

HALO_DBK_P, HALO_DBK_Q, HALO_ALF = 4, 8, 4      # UVGHIP_HALO_* (include/uvg266_hip.h), luma rows


class BandLayout:
    """Rows owned by `rank` of `nranks`, and the halo / gather lists over named planes."""

    def __init__(self, height, nranks, rank):
        L = _lib.load_library()
        p = _lib.BandPlan()
        rc = L.uvghip_band_plan(height, nranks, rank, ctypes.byref(p))
        if rc != 0:
            raise ValueError(f"uvghip_band_plan({height}, {nranks}, {rank}): {L.uvghip_last_error().decode()}")
        self.height, self.nranks, self.rank = height, nranks, rank
        self.ctu_rows, self.ctu_row0, self.ctu_row1 = p.ctu_rows, p.ctu_row0, p.ctu_row1
        self.y0, self.y1, self.up, self.down = p.y0, p.y1, p.up, p.down

    def owned(self, sub=1):
        """Owned rows of a plane subsampled by `sub` (1 luma, 2 chroma 4:2:0, 4 the per-4x4 tables)."""
        return self.y0 // sub, (self.y1 + sub - 1) // sub

    def _rows(self, t, lo, hi, sub):
        """Luma rows [lo, hi) clipped to the picture -> (tensor, row0, row1) of a plane subsampled by sub."""
        lo, hi = max(lo, 0), min(hi, self.height)
        return (t, lo // sub, (hi + sub - 1) // sub)

    def halo_deblock(self, y, u, v, scu):
        """After the vertical-edge pass: a band's horizontal-edge pass filters both of its boundary edges, so it needs
        the P side (4 luma rows) from above and the Q side (8 luma rows) from below, chroma half of that, and one row of
        per-4x4 side information on either side."""
        ops = []
        if self.up >= 0:
            a = self.y0
            ops.append((self.up,
                        [self._rows(y, a, a + HALO_DBK_Q, 1), self._rows(u, a, a + HALO_DBK_Q, 2), self._rows(v, a, a + HALO_DBK_Q, 2),
                         self._rows(scu, a, a + 4, 4)],
                        [self._rows(y, a - HALO_DBK_P, a, 1), self._rows(u, a - HALO_DBK_P, a, 2), self._rows(v, a - HALO_DBK_P, a, 2),
                         self._rows(scu, a - 4, a, 4)]))
        if self.down >= 0:
            b = self.y1
            ops.append((self.down,
                        [self._rows(y, b - HALO_DBK_P, b, 1), self._rows(u, b - HALO_DBK_P, b, 2), self._rows(v, b - HALO_DBK_P, b, 2),
                         self._rows(scu, b - 4, b, 4)],
                        [self._rows(y, b, b + HALO_DBK_Q, 1), self._rows(u, b, b + HALO_DBK_Q, 2), self._rows(v, b, b + HALO_DBK_Q, 2),
                         self._rows(scu, b, b + 4, 4)]))
        return ops

    def halo_search(self, y, u, v, scu, models, ctus_per_row):
        """The closed-loop CTU search of a band (uvghip_ctu_plan_create_rows) reads of the band above: its last line of the reconstruction
        (hor_buf_search), its last row of per-4x4 side information and the models the coder holds after its last row's FIRST CTU (the
        WPP context hand-over).  One direction: sent down after the band is searched, received from above before the search starts.
        y, u, v: reconstruction planes; scu: [rows of 4x4 units, bytes]; models: [CTUs, words] (row = CTU index)."""
        ops = []
        if self.up >= 0:
            a = self.y0
            k = (self.ctu_row0 - 1) * ctus_per_row
            ops.append((self.up, [], [self._rows(y, a - 1, a, 1), (u, a // 2 - 1, a // 2), (v, a // 2 - 1, a // 2), (scu, a // 4 - 1, a // 4), (models, k, k + 1)]))
        if self.down >= 0:
            b = self.y1
            k = (self.ctu_row1 - 1) * ctus_per_row
            ops.append((self.down, [self._rows(y, b - 1, b, 1), (u, b // 2 - 1, b // 2), (v, b // 2 - 1, b // 2), (scu, b // 4 - 1, b // 4), (models, k, k + 1)], []))
        return ops

    def halo_alf(self, y, u, v):
        """After SAO: ALF classification, filtering and statistics of a band read 3 rows of SAO output across either
        boundary (clamped at the virtual boundary, src/alf.h:32-33); 4 luma / 2 chroma rows are exchanged."""
        ops = []
        if self.up >= 0:
            a = self.y0
            ops.append((self.up, [self._rows(p, a, a + HALO_ALF, s) for p, s in ((y, 1), (u, 2), (v, 2))],
                        [self._rows(p, a - HALO_ALF, a, s) for p, s in ((y, 1), (u, 2), (v, 2))]))
        if self.down >= 0:
            b = self.y1
            ops.append((self.down, [self._rows(p, b - HALO_ALF, b, s) for p, s in ((y, 1), (u, 2), (v, 2))],
                        [self._rows(p, b, b + HALO_ALF, s) for p, s in ((y, 1), (u, 2), (v, 2))]))
        return ops

    def gather(self, y, u, v):
        """Every rank's reconstructed band to every other rank (the reference picture for inter prediction of the next
        frame is the whole picture on every GPU).  Bands are uneven, so this is a list of pairwise transfers."""
        ops = []
        for peer in range(self.nranks):
            if peer == self.rank:
                continue
            other = BandLayout(self.height, self.nranks, peer)
            ops.append((peer, [self._rows(p, self.y0, self.y1, s) for p, s in ((y, 1), (u, 2), (v, 2))],
                        [self._rows(p, other.y0, other.y1, s) for p, s in ((y, 1), (u, 2), (v, 2))]))
        return ops


def _nbytes(entry):
    t, r0, r1 = entry
    return (r1 - r0) * t.stride(0) * t.element_size()


def spec_bytes(ops):
    """(bytes sent, bytes received) by this rank for an exchange list."""
    return (sum(_nbytes(e) for _, s, _ in ops for e in s), sum(_nbytes(e) for _, _, r in ops for e in r))


def _ptr(entry):
    t, r0, _ = entry
    return t.data_ptr() + r0 * t.stride(0) * t.element_size()


class RcclTransport:
    """The product transport: RCCL through the C ABI.  Rendezvous (distributing the 128-byte unique id) goes through
    whatever the launcher offers; `bootstrap` is a callable rank0_bytes -> bytes on every rank (e.g. a
    torch.distributed broadcast over the launcher's store)."""

    def __init__(self, rank, nranks, bootstrap):
        self.L = _lib.load_library()
        self.rank, self.nranks = rank, nranks
        ident = ctypes.create_string_buffer(128)
        if rank == 0:
            _lib.check(self.L.uvghip_comm_unique_id(ident), "uvghip_comm_unique_id")
        raw = bootstrap(bytes(ident.raw))
        self.comm = ctypes.c_void_p()
        _lib.check(self.L.uvghip_comm_create(ctypes.create_string_buffer(raw, 128), nranks, rank, ctypes.byref(self.comm)),
                   "uvghip_comm_create")
        self._keep = []

    def compile(self, ops):
        """Exchange list -> (C array of uvghip_xfer_t, count): built once per picture slot, reused every step."""
        xs = []
        for peer, sends, recvs in ops:
            for i in range(max(len(sends), len(recvs))):
                x = _lib.Xfer()
                x.peer = peer
                if i < len(sends):
                    x.send, x.send_bytes = _ptr(sends[i]), _nbytes(sends[i])
                if i < len(recvs):
                    x.recv, x.recv_bytes = _ptr(recvs[i]), _nbytes(recvs[i])
                xs.append(x)
        arr = (_lib.Xfer * max(1, len(xs)))(*xs)
        self._keep.append(arr)
        return arr, len(xs)

    def launch_args(self, ops):
        """(fn, args) in the launch-plan convention fn(*args, stream)."""
        arr, n = self.compile(ops)
        return self.L.uvghip_comm_exchange, [self.comm, ctypes.cast(arr, ctypes.c_void_p), n]

    def exchange(self, ops, stream):
        fn, args = self.launch_args(ops)
        _lib.check(fn(*args, stream), "uvghip_comm_exchange")

    def allreduce_args(self, t):
        return self.L.uvghip_comm_allreduce_i64, [self.comm, t.data_ptr(), t.numel()]

    def close(self):
        if self.comm:
            self.L.uvghip_comm_destroy(self.comm)
            self.comm = ctypes.c_void_p()


class TorchTransport:
    """torch.distributed point-to-point (gloo on CPU tensors in the tests)."""

    def __init__(self, dist):
        self.dist = dist

    def exchange(self, ops, stream=None):
        d = self.dist
        p2p = []
        for peer, sends, recvs in ops:
            for t, r0, r1 in sends:
                p2p.append(d.P2POp(d.isend, t[r0:r1], peer))
            for t, r0, r1 in recvs:
                p2p.append(d.P2POp(d.irecv, t[r0:r1], peer))
        if p2p:
            for req in d.batch_isend_irecv(p2p):
                req.wait()

    def allreduce(self, t):
        self.dist.all_reduce(t)


def emulate(all_ops):
    """all_ops[r] = exchange list of emulated rank r (tensors of rank r).  Performs every receive as a row copy from the
    matching send of the peer, after checking that the two lists pair up (same order, same byte counts)."""
    for r, ops in enumerate(all_ops):
        for peer, _, recvs in ops:
            back = [o for o in all_ops[peer] if o[0] == r]
            assert len(back) == 1, f"rank {peer} has no exchange entry for rank {r}"
            sends = back[0][1]
            if not recvs:
                continue              # a one-directional exchange: this side only sends
            assert len(sends) == len(recvs), (r, peer)
            for (dt, d0, d1), (st, s0, s1) in zip(recvs, sends):
                assert d1 - d0 == s1 - s0 and (d0, d1) == (s0, s1), "halo rows keep their picture coordinates"
                dt[d0:d1].copy_(st[s0:s1])
