"""Host side of the wave codec: the reference's Packet / DataElement / filter surface,
one wave (n packets of one format string) at a time, over the C ABI of include/mrpdr.h.

PyTorch is plumbing only: device buffers, the current HIP stream, torch.distributed.
All arithmetic and byte movement happens in the HIP kernels of mrnet_amd/csrc.

Mirrors, for a wave instead of a packet:
  Fmt2Type / DataType         src/DataElement.C:205-297, include/mrnet/DataElement.h:27-45
  Packet ctor + encode        src/Packet.C:137-275, 54-85
  Packet(hdr, buf) + decode   src/Packet.C:277-295, 87-133
  Packet::unpack              src/Packet.C:329-353, 522-542 (returns -1 on a format mismatch)
  tfilter_Sum/Max/Min/Avg     src/FilterDefinitions.C:90-645
"""
import ctypes as C
import enum

import numpy as np
import torch

from . import capi


class DataType(enum.IntEnum):
    UNKNOWN_T = 0
    CHAR_T = 1
    UCHAR_T = 2
    CHAR_ARRAY_T = 3
    UCHAR_ARRAY_T = 4
    STRING_T = 5
    STRING_ARRAY_T = 6
    INT16_T = 7
    UINT16_T = 8
    INT16_ARRAY_T = 9
    UINT16_ARRAY_T = 10
    INT32_T = 11
    UINT32_T = 12
    INT32_ARRAY_T = 13
    UINT32_ARRAY_T = 14
    INT64_T = 15
    UINT64_T = 16
    INT64_ARRAY_T = 17
    UINT64_ARRAY_T = 18
    FLOAT_T = 19
    DOUBLE_T = 20
    FLOAT_ARRAY_T = 21
    DOUBLE_ARRAY_T = 22
    CHAR_LRG_ARRAY_T = 23
    UCHAR_LRG_ARRAY_T = 24
    STRING_LRG_ARRAY_T = 25
    INT16_LRG_ARRAY_T = 26
    UINT16_LRG_ARRAY_T = 27
    INT32_LRG_ARRAY_T = 28
    UINT32_LRG_ARRAY_T = 29
    INT64_LRG_ARRAY_T = 30
    UINT64_LRG_ARRAY_T = 31
    FLOAT_LRG_ARRAY_T = 32
    DOUBLE_LRG_ARRAY_T = 33


LITTLEENDIAN, BIGENDIAN = 0, 1
SCALAR, ARRAY, STRING, STRING_ARRAY = 0, 1, 2, 3
TFILTER_SUM, TFILTER_MIN, TFILTER_MAX, TFILTER_AVG = 0, 1, 2, 3      # operator codes of the C ABI
TFILTER_ARRAY_CONCAT = 6          # tfilter_PerfData's concatenating aggregation (host-side code only)
FILTER_OR, FILTER_AND = 4, 5      # bitwise (not built-in TFILTER_*: the BitsetOr / IntegerPercentiles examples)
SHAPE_FLAT, SHAPE_TREE, SHAPE_FLAT_SERIAL = 0, 1, 2
TREE_FANINS = (4, 2, 2, 2, 2, 2, 2, 4, 0)

USER_STRM_BASE_ID = (1 << 30) + 10      # src/Stream.C:31-42
FIRST_APPLICATION_TAG = 100             # include/mrnet/Types.h:20
UNKNOWN_RANK = 0xFFFFFFFF               # include/mrnet/Types.h (UnknownRank)

# device storage: torch has no arithmetic on unsigned >8 bit, so unsigned columns are
# held in the signed dtype of the same width (same bits)
_STORE = {1: torch.int8, 2: torch.int16, 4: torch.int32, 8: torch.int64}
_NP_ELEM = {
    "c": np.int8, "uc": np.uint8, "hd": np.int16, "uhd": np.uint16, "d": np.int32, "ud": np.uint32,
    "ld": np.int64, "uld": np.uint64, "f": np.float32, "lf": np.float64, "s": np.uint8,
}


class PdrError(RuntimeError):
    """A C-ABI call returned FALSE (the reference would raise ERR_PACKING)."""


def _check(ok, what):
    if not ok:
        raise PdrError("%s: %s" % (what, capi.last_error()))


def _ptr(t):
    return None if t is None else t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


class Format:
    """A compiled format string (tokenised once, not once per packet)."""

    def __init__(self, fmt):
        self._lib = capi.load()
        self.text = fmt
        self.handle = self._lib.mrpdr_fmt_compile(fmt.encode())
        if not self.handle:
            raise ValueError(capi.last_error())
        n = self._lib.mrpdr_fmt_nfields(self.handle)
        self.types = [DataType(self._lib.mrpdr_fmt_field_type(self.handle, i)) for i in range(n)]
        self.kinds = [self._lib.mrpdr_fmt_field_kind(self.handle, i) for i in range(n)]
        self.widths = [self._lib.mrpdr_fmt_field_width(self.handle, i) for i in range(n)]
        self.is_fixed = bool(self._lib.mrpdr_fmt_is_fixed(self.handle))
        self.fixed_bytes = int(self._lib.mrpdr_fmt_fixed_bytes(self.handle))
        self.header_bytes = int(self._lib.mrpdr_header_bytes(self.handle))
        self.tokens = [t for t in fmt.replace("%", " ").split()]

    def __len__(self):
        return len(self.types)

    def np_dtype(self, i):
        tok = self.tokens[i]
        tok = tok[1:] if tok[0] in "aA" else tok
        return np.dtype(_NP_ELEM[tok])

    def __del__(self):
        try:
            if self.handle:
                self._lib.mrpdr_fmt_free(self.handle)
                self.handle = None
        except Exception:
            pass


_TORCH_ELEM = {"c": torch.int8, "uc": torch.uint8, "hd": torch.int16, "uhd": torch.int16, "d": torch.int32,
               "ud": torch.int32, "ld": torch.int64, "uld": torch.int64, "f": torch.float32, "lf": torch.float64,
               "s": torch.uint8}


def torch_dtype(fmt, i):
    """Element dtype of field i as a torch view (unsigned wider than 8 bits: the signed dtype of the
    same width, same bits)."""
    tok = fmt.tokens[i]
    return _TORCH_ELEM[tok[1:] if tok[0] in "aA" else tok]


def Fmt2Type(token):
    """DataType of one format token, e.g. Fmt2Type('uld') (src/DataElement.C:205-297)."""
    try:
        f = Format("%" + token)
    except ValueError:
        return DataType.UNKNOWN_T
    return f.types[0] if len(f) == 1 else DataType.UNKNOWN_T


class Column:
    """One decoded field of a wave on the device (struct mrpdr_column)."""

    def __init__(self, values=None, counts=None, offs=None, capacity=None, sub_counts=None, sub_offs=None,
                 sub_capacity=None):
        self.values, self.counts, self.offs = values, counts, offs
        self.capacity = capacity if capacity is not None else (0 if values is None else values.numel())
        # string arrays ('%as'): counts/offs number the strings of every packet, sub_counts[j] is
        # strlen+1 of string j of the wave, values[sub_offs[j]:sub_offs[j+1]] its bytes
        self.sub_counts, self.sub_offs = sub_counts, sub_offs
        self.sub_capacity = sub_capacity if sub_capacity is not None else (
            0 if sub_counts is None else sub_counts.numel())

    def c_struct(self):
        return capi.Column(_ptr(self.values), _ptr(self.counts), _ptr(self.offs), self.capacity,
                           _ptr(self.sub_counts), _ptr(self.sub_offs), self.sub_capacity)


class WireWave:
    """Encoded data buffers of a wave: arena + offsets/lens, or a packed stride."""

    def __init__(self, arena, n, offsets=None, lens=None, stride=0):
        self.arena, self.n, self.offsets, self.lens, self.stride = arena, n, offsets, lens, stride

    def c_struct(self):
        return capi.Wire(_ptr(self.arena), _ptr(self.offsets), _ptr(self.lens), self.stride,
                         self.arena.numel())

    def packet_bytes(self, i):
        """Bytes of packet i as numpy (test / legacy-boundary helper: copies to the host)."""
        if self.offsets is not None:
            a = int(self.offsets[i].item())
        else:
            a = i * self.stride
        ln = int(self.lens[i].item()) if self.lens is not None else None
        return self.arena[a:a + ln].cpu().numpy() if ln is not None else self.arena[a:a + self.stride].cpu().numpy()


def _c_columns(cols):
    arr = (capi.Column * max(1, len(cols)))()
    for i, c in enumerate(cols):
        arr[i] = c.c_struct()
    return arr


class Codec:
    """A device context plus the wave operations (one per calling thread / stream)."""

    def __init__(self, device=None, max_packets=1 << 21):
        self._lib = capi.load()
        if not torch.cuda.is_available():
            raise PdrError("mrnet_amd needs a HIP device; there is no CPU fallback")
        self.device = torch.device("cuda", torch.cuda.current_device() if device is None else device)
        self.max_packets = max_packets
        self.handle = self._lib.mrpdr_ctx_create(self.device.index, max_packets)
        _check(self.handle, "mrpdr_ctx_create")

    def __del__(self):
        try:
            if self.handle:
                self._lib.mrpdr_ctx_destroy(self.handle)
                self.handle = None
        except Exception:
            pass

    # ---- buffers ---------------------------------------------------------------
    def empty(self, n, dtype):
        return torch.empty(int(n), dtype=dtype, device=self.device)

    def to_device(self, a):
        """numpy -> device tensor holding the same bytes (unsigned kept as signed)."""
        a = np.ascontiguousarray(a)
        if not a.flags.writeable:
            a = a.copy()
        if a.dtype.kind == "u" and a.dtype.itemsize > 1:
            a = a.view(np.dtype("i%d" % a.dtype.itemsize))
        return torch.from_numpy(a).to(self.device)

    @staticmethod
    def to_numpy(t, dtype=None):
        a = t.detach().cpu().numpy()
        return a if dtype is None else a.view(dtype)

    def columns_from_numpy(self, fmt, cols):
        """cols: per field either a numpy array (scalar) or (counts, values) for arrays and
        strings (string values hold every string's bytes NUL included), or
        (counts, sub_counts, values) for string arrays."""
        out = []
        for i, k in enumerate(fmt.kinds):
            if k == SCALAR:
                out.append(Column(self.to_device(np.asarray(cols[i], dtype=fmt.np_dtype(i)))))
            elif k == STRING_ARRAY:
                counts, sub_counts, values = cols[i]
                counts = np.asarray(counts, dtype=np.uint64)
                sub_counts = np.asarray(sub_counts, dtype=np.uint64)
                offs = np.zeros(len(counts) + 1, dtype=np.uint64)
                np.cumsum(counts, out=offs[1:])
                sub_offs = np.zeros(len(sub_counts) + 1, dtype=np.uint64)
                np.cumsum(sub_counts, out=sub_offs[1:])
                values = np.asarray(values, dtype=np.uint8)
                out.append(Column(self.to_device(values) if values.size else self.empty(1, torch.uint8),
                                  self.to_device(counts), self.to_device(offs), values.size,
                                  self.to_device(sub_counts) if sub_counts.size else self.empty(1, torch.int64),
                                  self.to_device(sub_offs), sub_counts.size))
            else:
                counts, values = cols[i]
                counts = np.asarray(counts, dtype=np.uint64)
                offs = np.zeros(len(counts) + 1, dtype=np.uint64)
                np.cumsum(counts, out=offs[1:])
                values = np.asarray(values, dtype=fmt.np_dtype(i))
                out.append(Column(self.to_device(values), self.to_device(counts), self.to_device(offs),
                                  capacity=values.size))
        return out

    def alloc_columns(self, fmt, n, capacities=None):
        out = []
        for i, k in enumerate(fmt.kinds):
            w = fmt.widths[i]
            if k == SCALAR:
                out.append(Column(self.empty(n, _STORE[w])))
            else:
                cap = int(capacities[i]) if capacities is not None else 0
                vals = self.empty(max(cap, 1), _STORE[w]) if cap else None
                out.append(Column(vals, self.empty(n, torch.int64), self.empty(n + 1, torch.int64), cap))
        return out

    # ---- sizing (pdr_sizeof over the wave) ---------------------------------------
    def sizeof(self, fmt, cols, n, align=1, phase=0, want_total=True, lens=None, offsets=None):
        lens = self.empty(n, torch.int64) if lens is None else lens
        offsets = self.empty(n + 1, torch.int64) if offsets is None else offsets
        total = C.c_uint64(0)
        ok = self._lib.mrpdr_wave_sizeof(self.handle, fmt.handle, _c_columns(cols), n, align, phase,
                                         _ptr(lens), _ptr(offsets),
                                         C.byref(total) if want_total else None, _stream())
        _check(ok, "mrpdr_wave_sizeof")
        return lens, offsets, (total.value if want_total else None)

    # ---- encode ---------------------------------------------------------------------
    def encode_into(self, fmt, cols, wire):
        w = wire.c_struct()
        ok = self._lib.mrpdr_wave_encode(self.handle, fmt.handle, _c_columns(cols), wire.n,
                                         C.byref(w), _stream())
        _check(ok, "mrpdr_wave_encode")

    def encode_begin(self, fmt, cols, wire):
        """Scalars, length prefixes and the plan of the payload streams (reads no array values)."""
        w = wire.c_struct()
        _check(self._lib.mrpdr_wave_encode_begin(self.handle, fmt.handle, _c_columns(cols), wire.n,
                                                 C.byref(w), _stream()), "mrpdr_wave_encode_begin")

    def encode_finish(self, fmt, cols, wire):
        w = wire.c_struct()
        _check(self._lib.mrpdr_wave_encode_finish(self.handle, fmt.handle, _c_columns(cols), wire.n,
                                                  C.byref(w), _stream()), "mrpdr_wave_encode_finish")

    def payload_phase(self, fmt, align):
        return int(self._lib.mrpdr_fmt_payload_phase(fmt.handle, align))

    def encode(self, fmt, cols, n, align=1, packed=None, phase=0):
        """Encode a wave.  Fixed formats default to the packed layout (no offsets array)."""
        if packed is None:
            packed = fmt.is_fixed
        if packed:
            if not fmt.is_fixed:
                raise PdrError("packed layout needs a scalar-only format")
            stride = (fmt.fixed_bytes + align - 1) // align * align
            wire = WireWave(torch.zeros(max(n * stride, 1), dtype=torch.uint8, device=self.device),
                            n, None, None, stride)
        else:
            lens, offsets, total = self.sizeof(fmt, cols, n, align, phase)
            wire = WireWave(torch.zeros(max(total, 1), dtype=torch.uint8, device=self.device),
                            n, offsets, lens, 0)
        self.encode_into(fmt, cols, wire)
        return wire

    # ---- decode ---------------------------------------------------------------------
    # sender_bo below: LITTLEENDIAN / BIGENDIAN for the whole wave, or a uint8 device tensor with
    # the byte order of every packet (as decode_headers returns it) for a wave of both kinds.
    def _decode_fn(self, name, sender_bo):
        if isinstance(sender_bo, torch.Tensor):
            if sender_bo.dtype != torch.uint8 or not sender_bo.is_contiguous():
                raise ValueError("per-packet byte orders must be a contiguous uint8 tensor")
            return getattr(self._lib, name.replace("decode", "decode_mixed")), _ptr(sender_bo)
        return getattr(self._lib, name), sender_bo

    def decode_begin(self, fmt, wire, cols, ok, sender_bo=LITTLEENDIAN, want_totals=True):
        w = wire.c_struct()
        totals = (C.c_uint64 * max(1, 2 * len(fmt)))()
        carr = _c_columns(cols)
        fn, bo = self._decode_fn("mrpdr_wave_decode_begin", sender_bo)
        _check(fn(self.handle, fmt.handle, C.byref(w), wire.n, bo, carr, _ptr(ok),
                  totals if want_totals else None, _stream()), "mrpdr_wave_decode_begin")
        return [totals[i] for i in range(2 * len(fmt))] if want_totals else None

    def decode_finish(self, fmt, wire, cols, sender_bo=LITTLEENDIAN):
        w = wire.c_struct()
        fn, bo = self._decode_fn("mrpdr_wave_decode_finish", sender_bo)
        _check(fn(self.handle, fmt.handle, C.byref(w), wire.n, bo, _c_columns(cols), _stream()),
               "mrpdr_wave_decode_finish")

    def decode_into(self, fmt, wire, cols, ok, sender_bo=LITTLEENDIAN):
        """Columns already sized by the caller: no host synchronisation."""
        w = wire.c_struct()
        fn, bo = self._decode_fn("mrpdr_wave_decode", sender_bo)
        _check(fn(self.handle, fmt.handle, C.byref(w), wire.n, bo, _c_columns(cols), _ptr(ok), _stream()),
               "mrpdr_wave_decode")

    def decode_filter_into(self, fmt, wire, cols, ok, op, shape=SHAPE_FLAT, sender_bo=LITTLEENDIAN,
                           result=None, result_ok=None):
        """Decode a scalar-only wave and fold its first field in one call (a tree node's work on
        an upstream wave): columns and ok as decode_into, (result, result_ok) as filter_columns."""
        result = self.empty(16, torch.uint8) if result is None else result
        result_ok = self.empty(1, torch.int32) if result_ok is None else result_ok
        w = wire.c_struct()
        _check(self._lib.mrpdr_wave_decode_filter(self.handle, fmt.handle, C.byref(w), wire.n, sender_bo,
                                                  _c_columns(cols), _ptr(ok), op, shape, _ptr(result),
                                                  _ptr(result_ok), _stream()), "mrpdr_wave_decode_filter")
        return result, result_ok

    # ---- calls bound once -------------------------------------------------------------------
    # A node that handles wave after wave of one stream calls the same entry points with the same
    # buffers every time.  bind_* converts the arguments to their C form once (structs, raw device
    # pointers, the HIP stream that is current AT BIND TIME) and returns a function that only makes
    # the call.  The function holds on to the objects its pointers come from (wire, columns, ok,
    # result tensors: `call.owners`), so the buffers cannot be freed under it; it does not follow a
    # tensor that the caller REPLACES in one of those objects afterwards, nor a change of the
    # current torch stream: bind again after either.
    def _bound(self, name, owners, *args):
        fn = getattr(self._lib, name)

        def call():
            if not fn(*args):
                raise PdrError("%s: %s" % (name, capi.last_error()))
        call.keep = args                  # the ctypes structures must outlive the closure's calls
        call.owners = owners              # ... and so must the tensors their pointers point into
        return call

    @staticmethod
    def _tensors_of(*things):
        """The tensors behind wires, columns and plain tensors (what a bound call must keep alive)."""
        out = []
        for x in things:
            if x is None:
                continue
            if isinstance(x, torch.Tensor):
                out.append(x)
            elif isinstance(x, WireWave):
                out += [t for t in (x.arena, x.offsets, x.lens) if t is not None]
            elif isinstance(x, Column):
                out += [t for t in (x.values, x.counts, x.offs, x.sub_counts, x.sub_offs) if t is not None]
            elif isinstance(x, (list, tuple)):
                out += Codec._tensors_of(*x)
            else:
                out.append(x)             # (a Format: its handle is an argument too)
        return out

    def bind_decode(self, fmt, wire, cols, ok, sender_bo=LITTLEENDIAN):
        return self._bound("mrpdr_wave_decode", self._tensors_of(fmt, wire, cols, ok),
                           self.handle, fmt.handle, C.byref(wire.c_struct()), wire.n,
                           sender_bo, _c_columns(cols), _ptr(ok), _stream())

    def bind_encode(self, fmt, cols, wire):
        return self._bound("mrpdr_wave_encode", self._tensors_of(fmt, wire, cols),
                           self.handle, fmt.handle, _c_columns(cols), wire.n,
                           C.byref(wire.c_struct()), _stream())

    def bind_decode_filter(self, fmt, wire, cols, ok, op, shape, result, result_ok, sender_bo=LITTLEENDIAN):
        return self._bound("mrpdr_wave_decode_filter", self._tensors_of(fmt, wire, cols, ok, result, result_ok),
                           self.handle, fmt.handle, C.byref(wire.c_struct()), wire.n,
                           sender_bo, _c_columns(cols), _ptr(ok), op, shape, _ptr(result), _ptr(result_ok),
                           _stream())

    def bind_filter(self, op, dtype, values_ptr, value_stride, counts_ptr, count_stride, n, swap, shape,
                    result, result_ok, owners=()):
        """values_ptr / counts_ptr are raw device addresses: pass the tensors they point into as
        `owners` to have the bound call keep them alive."""
        return self._bound("mrpdr_wave_filter", self._tensors_of(result, result_ok, list(owners)),
                           self.handle, op, int(dtype), shape, values_ptr, value_stride,
                           counts_ptr, count_stride, n, swap, _ptr(result), _ptr(result_ok), _stream())

    def decode(self, fmt, wire, sender_bo=LITTLEENDIAN):
        """Decode a wave; returns (columns, ok).  Array columns are allocated from the
        totals the walk reports, so this synchronises once for variable formats."""
        n = wire.n
        cols = self.alloc_columns(fmt, n)
        ok = self.empty(n, torch.int32)
        totals = self.decode_begin(fmt, wire, cols, ok, sender_bo, want_totals=not fmt.is_fixed)
        if not fmt.is_fixed:
            for i, k in enumerate(fmt.kinds):
                if k == STRING_ARRAY:
                    ns, nb = int(totals[i]), int(totals[len(fmt) + i])
                    cols[i].capacity, cols[i].sub_capacity = nb, ns
                    cols[i].values = self.empty(max(nb, 1), torch.int8)
                    cols[i].sub_counts = self.empty(max(ns, 1), torch.int64)
                    cols[i].sub_offs = self.empty(ns + 1, torch.int64)
                elif k != SCALAR:
                    cols[i].capacity = int(totals[i])
                    cols[i].values = self.empty(max(int(totals[i]), 1), _STORE[fmt.widths[i]])
            self.decode_finish(fmt, wire, cols, sender_bo)
        return cols, ok

    def status(self):
        s = C.c_uint32(0)
        _check(self._lib.mrpdr_ctx_status(self.handle, _stream(), C.byref(s)), "mrpdr_ctx_status")
        return s.value

    def perf_enable(self, on=1):
        """on: 0/False = off, k = bracket every k-th payload launch of each direction with events."""
        _check(self._lib.mrpdr_ctx_perf_enable(self.handle, int(on)), "mrpdr_ctx_perf_enable")

    def perf_collect(self):
        """{'decode': (ms_sum, launches), 'encode': (ms_sum, launches)} of the payload kernels
        since the last collect (HIP events on the stream the kernels ran on)."""
        ms = (C.c_double * 2)()
        cnt = (C.c_uint64 * 2)()
        _check(self._lib.mrpdr_ctx_perf_collect(self.handle, _stream(), ms, cnt), "mrpdr_ctx_perf_collect")
        return {"decode": (ms[0], cnt[0]), "encode": (ms[1], cnt[1])}

    def plain_copy_ms(self, dst, src, reps=20):
        """Average launch time of the calibration copy kernel (src -> dst, same byte count)."""
        ms = C.c_double(0.0)
        nbytes = src.numel() * src.element_size()
        _check(self._lib.mrpdr_ctx_perf_plain_copy(self.handle, _ptr(dst), _ptr(src), nbytes, reps,
                                                   C.byref(ms), _stream()), "mrpdr_ctx_perf_plain_copy")
        return ms.value

    # ---- headers ---------------------------------------------------------------------
    def encode_headers(self, fmt, n, stream_id=USER_STRM_BASE_ID, tag=FIRST_APPLICATION_TAG,
                       src_rank=UNKNOWN_RANK, stream_ids=None, tags=None, src_ranks=None,
                       stride=None, out=None, out_ptr=None, offsets=None, dest_ranks=None):
        """Headers of the wave.  Default: a fresh buffer, header i at i*stride.  With out_ptr
        (a device address) and offsets, header i goes to out_ptr + offsets[i].  dest_ranks: the
        destination list (device int32/uint32 tensor) every packet of a downstream wave carries."""
        ndest = 0 if dest_ranks is None else dest_ranks.numel()
        stride = fmt.header_bytes + 4 * ndest if stride is None else stride
        if out_ptr is None:
            hdrs = out if out is not None else torch.zeros(max(n * stride, 1), dtype=torch.uint8,
                                                           device=self.device)
            out_ptr = hdrs.data_ptr()
        else:
            hdrs = out
        r = self._lib.mrpdr_wave_encode_headers(self.handle, fmt.handle, n, stream_id, _ptr(stream_ids),
                                                tag, _ptr(tags), src_rank, _ptr(src_ranks),
                                                _ptr(dest_ranks), ndest,
                                                out_ptr, _ptr(offsets), stride, _stream())
        _check(r, "mrpdr_wave_encode_headers")
        return hdrs, stride

    def decode_headers(self, fmt, n, hdrs, stride=0, offsets=None, lens=None, max_dest=0):
        """Header fields of the wave; with max_dest, also the first max_dest destination ranks of
        every packet (out['dest_ranks'], n x max_dest)."""
        dests = self.empty(n * max_dest, torch.int32) if max_dest else None
        out = {
            "stream_id": self.empty(n, torch.int32), "tag": self.empty(n, torch.int32),
            "src_rank": self.empty(n, torch.int32), "byteorder": self.empty(n, torch.uint8),
            "ndest": self.empty(n, torch.int64), "ok": self.empty(n, torch.int32),
        }
        r = self._lib.mrpdr_wave_decode_headers(self.handle, fmt.handle, n, _ptr(hdrs),
                                                0 if hdrs is None else hdrs.numel(), _ptr(offsets),
                                                _ptr(lens), stride, _ptr(out["stream_id"]),
                                                _ptr(out["tag"]), _ptr(out["src_rank"]),
                                                _ptr(out["byteorder"]), _ptr(out["ndest"]),
                                                _ptr(dests), max_dest, _ptr(out["ok"]), _stream())
        _check(r, "mrpdr_wave_decode_headers")
        if max_dest:
            out["dest_ranks"] = dests.view(n, max_dest)
        return out

    # ---- message framing (src/Message.C:48-393) -------------------------------------------
    def message_build(self, fmt, cols, n, stream_id=USER_STRM_BASE_ID, tag=FIRST_APPLICATION_TAG,
                      src_rank=UNKNOWN_RANK, src_ranks=None):
        """One Message holding the wave, encoded in place: returns the uint8 device tensor that a
        single writev would put on the link.  (One host read: the payload size, to allocate it.)"""
        h = fmt.header_bytes
        lens, offsets, payload = self.sizeof(fmt, cols, n, 1, h)
        pre = int(self._lib.mrpdr_message_preamble_bytes(n))
        msg = self.empty(pre + payload, torch.uint8)          # every byte of it is written below
        self.message_build_into(fmt, cols, n, lens, offsets, msg, stream_id, tag, src_rank, src_ranks)
        return msg

    def message_build_into(self, fmt, cols, n, lens, offsets, msg, stream_id=USER_STRM_BASE_ID,
                           tag=FIRST_APPLICATION_TAG, src_rank=UNKNOWN_RANK, src_ranks=None):
        """Frame (preamble + headers, one launch) and data of the wave into msg; lens / offsets from
        sizeof(fmt, cols, n, 1, fmt.header_bytes).  No host synchronisation."""
        pre = int(self._lib.mrpdr_message_preamble_bytes(n))
        _check(self._lib.mrpdr_message_write_frame(self.handle, fmt.handle, n, _ptr(lens), _ptr(offsets),
                                                   stream_id, None, tag, None, src_rank, _ptr(src_ranks),
                                                   _ptr(msg), _stream()), "mrpdr_message_write_frame")
        self.encode_into(fmt, cols, WireWave(msg[pre:], n, offsets, lens, 0))

    def message_workspace(self, n, max_dest=0):
        """The outputs of message_parse for waves of up to n packets, allocated once."""
        ws = {
            "hdr_offs": self.empty(n + 1, torch.int64), "hdr_lens": self.empty(n, torch.int64),
            "buf_offs": self.empty(n + 1, torch.int64), "buf_lens": self.empty(n, torch.int64),
            "stream_id": self.empty(n, torch.int32), "tag": self.empty(n, torch.int32),
            "src_rank": self.empty(n, torch.int32), "byteorder": self.empty(n, torch.uint8),
            "ndest": self.empty(n, torch.int64), "ok": self.empty(n, torch.int32),
        }
        if max_dest:
            ws["dest_ranks"] = self.empty(n * max_dest, torch.int32).view(n, max_dest)
        return ws

    def message_parse(self, fmt, msg, n=None, check_length=True, workspace=None, max_dest=0):
        """Message -> (n, header fields, WireWave of the data buffers).  Message::recv reads the
        5-byte count from the socket before it knows how much more to receive (src/Message.C:71-86),
        so a caller that received the message itself passes that count as `n`; without it the
        count is read back from the device buffer here (one synchronisation).  check_length=False
        skips the host-side comparison of the size vector's total with the buffer length (another
        synchronisation): headers and data buffers that do not lie inside the buffer then simply
        fail as packets (ok = 0), nothing outside the buffer is touched.  workspace: the outputs
        (message_workspace), for a caller that parses message after message."""
        if n is None:
            head = msg[:5].cpu().numpy()
            bo = int(head[0])
            n = int(head[1:5].view(">u4" if bo == BIGENDIAN else "<u4")[0])
        pre = int(self._lib.mrpdr_message_preamble_bytes(n))
        if pre > msg.numel():
            raise PdrError("message shorter than its size vector")
        ws = self.message_workspace(n, max_dest) if workspace is None else workspace
        dests = ws.get("dest_ranks")
        r = self._lib.mrpdr_message_parse(self.handle, fmt.handle, _ptr(msg), msg.numel(), n,
                                          _ptr(ws["hdr_offs"]), _ptr(ws["hdr_lens"]), _ptr(ws["buf_offs"]),
                                          _ptr(ws["buf_lens"]), _ptr(ws["stream_id"]), _ptr(ws["tag"]),
                                          _ptr(ws["src_rank"]), _ptr(ws["byteorder"]), _ptr(ws["ndest"]),
                                          _ptr(dests), 0 if dests is None else dests.shape[1], _ptr(ws["ok"]),
                                          _stream())
        _check(r, "mrpdr_message_parse")
        if check_length:
            total = int(ws["hdr_offs"][n].item()) & (2**64 - 1)          # (a uint64 kept in an int64 tensor)
            if pre + total > msg.numel():
                raise PdrError("message shorter than its size vector announces")
            payload = msg[pre:pre + total]
        else:
            payload = msg[pre:]
        hdr = {k: ws[k][:n] for k in ("stream_id", "tag", "src_rank", "byteorder", "ndest", "ok")}
        if dests is not None:
            hdr["dest_ranks"] = dests[:n]
        return n, hdr, WireWave(payload, n, ws["buf_offs"], ws["buf_lens"], 0)

    # ---- filters -----------------------------------------------------------------------
    def filter_raw(self, op, dtype, values_ptr, value_stride, counts_ptr, count_stride, n, swap=0,
                   shape=SHAPE_FLAT, result=None, result_ok=None):
        if result is None:
            result = self.empty(16, torch.uint8)
        if result_ok is None:
            result_ok = self.empty(1, torch.int32)
        if isinstance(swap, torch.Tensor):        # one sender byte order per packet (uint8 device tensor)
            r = self._lib.mrpdr_wave_filter_mixed(self.handle, op, int(dtype), shape, values_ptr, value_stride,
                                                  counts_ptr, count_stride, n, _ptr(swap), _ptr(result),
                                                  _ptr(result_ok), _stream())
        else:
            r = self._lib.mrpdr_wave_filter(self.handle, op, int(dtype), shape, values_ptr, value_stride,
                                            counts_ptr, count_stride, n, swap, _ptr(result),
                                            _ptr(result_ok), _stream())
        _check(r, "mrpdr_wave_filter")
        return result, result_ok

    # ---- one filter call over a wave held in ranges (several GPUs, or several calls) ----------
    def is_chained(self, op, dtype):
        """True for float / double SUM and AVG: the chain of roundings has to be continued from
        range to range (front = the node the range before left), not re-associated."""
        return bool(self._lib.mrpdr_filter_is_chained(op, int(dtype)))

    def filter_partial_raw(self, op, dtype, values_ptr, value_stride, counts_ptr, count_stride, n, swap=0,
                           front=None, node=None):
        """The 16-byte node of a packet range (struct mrpdr_node as a uint8[16] device tensor);
        front: the node of the packets in front of the range, or None."""
        node = self.empty(16, torch.uint8) if node is None else node
        if isinstance(swap, torch.Tensor):        # one sender byte order per packet (uint8 device tensor)
            _check(self._lib.mrpdr_wave_filter_partial_mixed(self.handle, op, int(dtype), values_ptr, value_stride, counts_ptr,
                                                             count_stride, n, _ptr(swap), _ptr(front), _ptr(node), _stream()),
                   "mrpdr_wave_filter_partial_mixed")
            return node
        _check(self._lib.mrpdr_wave_filter_partial(self.handle, op, int(dtype), values_ptr, value_stride, counts_ptr,
                                                   count_stride, n, swap, _ptr(front), _ptr(node), _stream()),
               "mrpdr_wave_filter_partial")
        return node

    def filter_partial_wire(self, op, fmt, wire, sender_bo=LITTLEENDIAN, front=None, node=None):
        """filter_partial_raw straight on the wire bytes of a packed fixed-format wave (as filter_wire)."""
        want = 2 if op == TFILTER_AVG else 1
        if len(fmt) != want or not fmt.is_fixed or (want == 2 and fmt.types[1] != DataType.INT32_T):
            raise PdrError("filter format must be '%T'" + (" %d'" if want == 2 else "'"))
        if wire.offsets is not None:
            raise PdrError("filter_partial_wire needs the packed layout")
        base = wire.arena.data_ptr()
        return self.filter_partial_raw(op, fmt.types[0], base, wire.stride, base + fmt.widths[0], wire.stride, wire.n,
                                       int(sender_bo != LITTLEENDIAN), front, node)

    def filter_combine(self, op, dtype, nodes, result=None, result_ok=None):
        """nodes: uint8[g, 16] (or uint8[16 * g]) device tensor, range order -> (result, result_ok) of
        the one reference call over the whole wave."""
        result = self.empty(16, torch.uint8) if result is None else result
        result_ok = self.empty(1, torch.int32) if result_ok is None else result_ok
        nodes = nodes.contiguous()
        _check(self._lib.mrpdr_wave_filter_combine(self.handle, op, int(dtype), _ptr(nodes), nodes.numel() // 16,
                                                   _ptr(result), _ptr(result_ok), _stream()),
               "mrpdr_wave_filter_combine")
        return result, result_ok

    def filter_stats_raw(self, dtype, values_ptr, value_stride, counts_ptr, count_stride, n, swap=0,
                         results=None, results_ok=None):
        """Min, Max and Avg of a '<T> %d' wave in one pass: (uint8[48], int32[3])."""
        results = self.empty(48, torch.uint8) if results is None else results
        results_ok = self.empty(3, torch.int32) if results_ok is None else results_ok
        r = self._lib.mrpdr_wave_filter_stats(self.handle, int(dtype), values_ptr, value_stride, counts_ptr,
                                              count_stride, n, swap, _ptr(results), _ptr(results_ok), _stream())
        _check(r, "mrpdr_wave_filter_stats")
        return results, results_ok

    # ---- array filters -------------------------------------------------------------------
    def array_concat(self, col, n):
        """tfilter_ArrayConcat on a decoded array field: the Column of the ONE output packet
        (it shares the input's values; only its count and offsets are new)."""
        out = Column(col.values, self.empty(1, torch.int64), self.empty(2, torch.int64), col.capacity,
                     col.sub_counts, col.sub_offs, col.sub_capacity)
        a, b = col.c_struct(), out.c_struct()
        _check(self._lib.mrpdr_wave_array_concat(self.handle, C.byref(a), n, C.byref(b), _stream()),
               "mrpdr_wave_array_concat")
        return out

    def array_concat_masked(self, col, width, mask_counts, n):
        """Concatenation of the arrays of the packets whose mask count is non-zero (a copy)."""
        cap = max(int(col.capacity), 1)
        out = Column(self.empty(cap * width, torch.int8), self.empty(1, torch.int64), self.empty(2, torch.int64), cap)
        a, b = col.c_struct(), out.c_struct()
        _check(self._lib.mrpdr_wave_array_concat_masked(self.handle, C.byref(a), width, _ptr(mask_counts), n,
                                                        C.byref(b), _stream()), "mrpdr_wave_array_concat_masked")
        return out

    def filter_perfdata(self, op, dtype, ranks, data, n, group=0, out_cap=None):
        """tfilter_PerfData (SUM/MIN/MAX/AVG) over the rank and data columns of a decoded wave.
        Returns (out_data, out_len, out_rank) device tensors; out_cap defaults to the longest
        possible result (one host read of the data counts)."""
        if out_cap is None:
            out_cap = int(data.counts.max().item()) if n else 0
        out = self.empty(max(out_cap, 1), torch.int64)
        out_len, out_rank = self.empty(1, torch.int64), self.empty(1, torch.int32)
        need = int(self._lib.mrpdr_perfdata_scratch_bytes(n, group, out_cap))
        scratch = self.empty(max(need, 16), torch.uint8) if need else None
        a, b = ranks.c_struct(), data.c_struct()
        r = self._lib.mrpdr_wave_perfdata(self.handle, op, int(dtype), C.byref(a), C.byref(b), n, group,
                                          _ptr(out), out_cap, _ptr(out_len), _ptr(out_rank),
                                          _ptr(scratch), need, _stream())
        _check(r, "mrpdr_wave_perfdata")
        return out, out_len, out_rank

    def filter_perfdata_levels(self, op, dtype, ranks, data, n, fanins, out_cap=None, outputs=None, scratch=None):
        """tfilter_PerfData as a declared tree: level l folds groups of fanins[l] packets of the level
        below (0 = all), one filter invocation per group.  Returns (out_data, out_len, out_rank);
        outputs / scratch: those three tensors and the scratch of an earlier call, for a caller that
        filters wave after wave."""
        if out_cap is None:
            out_cap = int(data.counts.max().item()) if n else 0
        fan = (C.c_uint64 * len(fanins))(*[int(f) for f in fanins])
        need = int(self._lib.mrpdr_perfdata_levels_scratch_bytes(n, fan, len(fanins), out_cap))
        if outputs is None:
            outputs = (self.empty(max(out_cap, 1), torch.int64), self.empty(1, torch.int64), self.empty(1, torch.int32))
        out, out_len, out_rank = outputs
        if scratch is None or scratch.numel() < need:
            scratch = self.empty(max(need, 16), torch.uint8) if need else None
        a, b = ranks.c_struct(), data.c_struct()
        _check(self._lib.mrpdr_wave_perfdata_levels(self.handle, op, int(dtype), C.byref(a), C.byref(b), n, fan, len(fanins),
                                                    _ptr(out), out_cap, _ptr(out_len), _ptr(out_rank), _ptr(scratch), need,
                                                    _stream()), "mrpdr_wave_perfdata_levels")
        return out, out_len, out_rank

    # ---- one PerfData invocation over a wave held in ranges -----------------------------------
    def perfdata_is_chained(self, op, dtype):
        return bool(self._lib.mrpdr_perfdata_is_chained(op, int(dtype)))

    def perfdata_state_bytes(self, cap):
        return int(self._lib.mrpdr_perfdata_state_bytes(cap))

    def perfdata_partial(self, op, dtype, ranks, data, n, cap, front=None, state=None):
        """The state (struct mrpdr_perf_state + cap accumulators, a uint8 device tensor) of a packet
        range; front: the state of the packets in front (doubles only)."""
        state = self.empty(self.perfdata_state_bytes(cap), torch.uint8) if state is None else state
        need = int(self._lib.mrpdr_perfdata_scratch_bytes(n, 0, cap))
        scratch = self.empty(max(need, 16), torch.uint8) if need else None
        a, b = ranks.c_struct(), data.c_struct()
        _check(self._lib.mrpdr_wave_perfdata_partial(self.handle, op, int(dtype), C.byref(a), C.byref(b), n, cap, _ptr(front),
                                                     _ptr(state), _ptr(scratch), need, _stream()), "mrpdr_wave_perfdata_partial")
        return state

    def perfdata_combine(self, op, dtype, states, cap, out_cap=None):
        """states: uint8[g, state_bytes] device tensor in range order -> (out_data, out_len, out_rank)."""
        out_cap = cap if out_cap is None else out_cap
        states = states.contiguous()
        g = states.numel() // self.perfdata_state_bytes(cap)
        out = self.empty(max(out_cap, 1), torch.int64)
        out_len, out_rank = self.empty(1, torch.int64), self.empty(1, torch.int32)
        _check(self._lib.mrpdr_wave_perfdata_combine(self.handle, op, int(dtype), _ptr(states), g, cap, _ptr(out), out_cap,
                                                     _ptr(out_len), _ptr(out_rank), _stream()), "mrpdr_wave_perfdata_combine")
        return out, out_len, out_rank

    def int_eq_class(self, vals, memcnts, mems, n):
        """tfilter_IntEqClass over the three decoded '%aud' columns of a wave: the columns of the
        one output packet (values, members per class, members).  Outputs are sized by the inputs
        (there cannot be more classes or members than came in)."""
        kcap, mcap = max(int(vals.capacity), 1), max(int(mems.capacity), 1)
        outs = [Column(self.empty(cap, torch.int32), self.empty(1, torch.int64), self.empty(2, torch.int64), cap)
                for cap in (kcap, kcap, mcap)]
        need = int(self._lib.mrpdr_eqclass_scratch_bytes(int(vals.capacity)))
        scratch = self.empty(need, torch.uint8)
        cs = [x.c_struct() for x in (vals, memcnts, mems)] + [x.c_struct() for x in outs]
        r = self._lib.mrpdr_wave_int_eq_class(self.handle, C.byref(cs[0]), C.byref(cs[1]), C.byref(cs[2]), n,
                                              C.byref(cs[3]), C.byref(cs[4]), C.byref(cs[5]),
                                              _ptr(scratch), need, _stream())
        _check(r, "mrpdr_wave_int_eq_class")
        return outs

    def filter_columns(self, op, dtype, values, counts=None, shape=SHAPE_FLAT):
        """Filter over decoded columns: values = T[n] tensor, counts = int32[n] (AVG)."""
        n = values.numel()
        return self.filter_raw(op, dtype, _ptr(values), values.element_size(),
                               _ptr(counts), 4, n, 0, shape)

    def filter_wire(self, op, fmt, wire, sender_bo=LITTLEENDIAN, shape=SHAPE_FLAT):
        """Fused decode + filter straight on the wire bytes of a packed fixed-format wave
        ('%T' for SUM/MIN/MAX, '%T %d' for AVG, as the built-in filters require)."""
        want = 2 if op == TFILTER_AVG else 1
        if len(fmt) != want or not fmt.is_fixed or (want == 2 and fmt.types[1] != DataType.INT32_T):
            raise PdrError("filter format must be '%T'" + (" %d'" if want == 2 else "'"))
        if wire.offsets is not None:
            raise PdrError("filter_wire needs the packed layout")
        base = wire.arena.data_ptr()
        w = fmt.widths[0]
        swap = sender_bo if isinstance(sender_bo, torch.Tensor) else int(sender_bo != LITTLEENDIAN)
        return self.filter_raw(op, fmt.types[0], base, wire.stride, base + w, wire.stride, wire.n, swap, shape)


def result_value(result, result_ok, fmt_or_dtype, op):
    """Host view of a filter result: (value, count or None), or None when absent."""
    if int(result_ok.item()) == 0:
        return None
    raw = result.cpu().numpy()
    dt = fmt_or_dtype if isinstance(fmt_or_dtype, np.dtype) else np.dtype(fmt_or_dtype)
    v = raw[:dt.itemsize].view(dt)[0]
    c = int(raw[dt.itemsize:dt.itemsize + 4].view(np.int32)[0]) if op == TFILTER_AVG else None
    return v, c
