"""`Packet` and the built-in transformation filters of the reference, for a wave at a time.

A PacketWave is what `std::vector<PacketPtr>` is to a filter in the reference
(src/Filter.h:64-68): n packets that share a stream id, a tag and a format string.  Method
names, argument meaning and error behaviour follow include/mrnet/Packet.h:41-118 and
src/Packet.C:

  * constructing from values encodes the data at once (src/Packet.C:137-165 -> encode_pdr_data);
    the header is encoded lazily, when it is first asked for (src/Message.C:288-292)
  * constructing from received buffers decodes the header at once and the data lazily, on the
    first unpack / [] (src/Packet.C:277-295, 343-367)
  * unpack(fmt) returns -1 when fmt differs from the packet's format string (strcmp,
    src/Packet.C:527-532)
  * a decode failure sets has_Error() and ok[i] = 0 for the packets concerned
    (error(ERR_PACKING), src/Packet.C:122-124)
  * tfilter_Sum / Max / Min / Avg return one packet with the stream id and tag of the first
    input packet and the same format (src/FilterDefinitions.C:151-221)
  * tfilter_ArrayConcat returns one packet whose array is the arrays of all inputs back to back
    (src/FilterDefinitions.C:649-810); tfilter_PerfData folds "%ad %ad %a{uld,ld,lf}" packets
    element by element (src/FilterDefinitions.C:904-1239)
"""
import numpy as np
import torch

from . import wave as W


class PacketWave:
    def __init__(self, codec, stream_id, tag, fmt, columns, n=None, src_rank=W.UNKNOWN_RANK,
                 src_ranks=None, align=1, phase=0):
        """API constructor: packets from per-field columns (numpy or device Columns)."""
        self.codec = codec
        self.fmt = fmt if isinstance(fmt, W.Format) else W.Format(fmt)
        self.stream_id, self.tag, self.src_rank, self.src_ranks = stream_id, tag, src_rank, src_ranks
        self.inlet_rank = W.UNKNOWN_RANK
        self._error = False
        if columns and not isinstance(columns[0], W.Column):
            columns = codec.columns_from_numpy(self.fmt, columns)
        self.columns = list(columns)
        if n is None:
            first = self.columns[0] if self.columns else None
            n = 0 if first is None else (first.counts.numel() if first.counts is not None
                                         else first.values.numel())
        self.n = n
        self.ok = None
        self._decoded = True
        self._hdrs = None
        self.byteorder = W.LITTLEENDIAN
        self.wire = codec.encode(self.fmt, self.columns, n, align=align, phase=phase)   # encode_pdr_data
        if not self.fmt.is_fixed and codec.status():
            self._error = True      # e.g. a NULL string: error(ERR_PACKING, ...), src/Packet.C:70-72

    @classmethod
    def from_buffers(cls, codec, fmt, n, hdrs, hdr_stride, wire, inlet_rank=W.UNKNOWN_RANK):
        """Receive-side constructor: Packet(hdr_len, hdr, buf_len, buf, inlet_rank)."""
        self = cls.__new__(cls)
        self.codec = codec
        self.fmt = fmt if isinstance(fmt, W.Format) else W.Format(fmt)
        self.n, self.wire, self.inlet_rank = n, wire, inlet_rank
        self._hdrs, self._hdr_stride = hdrs, hdr_stride
        self.columns, self.ok, self._decoded, self._error = None, None, False, False
        h = codec.decode_headers(self.fmt, n, hdrs, stride=hdr_stride)            # decode_pdr_header
        self.header = h
        self.src_ranks = h["src_rank"]
        self.src_rank = W.UNKNOWN_RANK
        if n:
            self.stream_id = int(h["stream_id"][0].item()) & 0xFFFFFFFF
            self.tag = int(h["tag"][0].item())
            bo = h["byteorder"]
            self.byteorder = int(bo[0].item())
            if not bool((bo == bo[0]).all()):
                self.byteorder = bo         # children of both kinds: every packet keeps its own order
            if not bool(h["ok"].all()):
                self._error = True          # a packet of another format in the wave
        else:
            self.stream_id, self.tag, self.byteorder = 0xFFFFFFFF, -1, W.LITTLEENDIAN
        return self

    @classmethod
    def from_message(cls, codec, fmt, msg, inlet_rank=W.UNKNOWN_RANK):
        """Message::recv (src/Message.C:48-199): the wave inside one message (uint8 device tensor),
        headers and data decoded in place."""
        self = cls.__new__(cls)
        self.codec = codec
        self.fmt = fmt if isinstance(fmt, W.Format) else W.Format(fmt)
        n, h, wire = codec.message_parse(self.fmt, msg)
        self.n, self.wire, self.inlet_rank = n, wire, inlet_rank
        self._hdrs, self._hdr_stride = None, 0          # they lie between the data buffers
        self.columns, self.ok, self._decoded, self._error = None, None, False, False
        self.header, self.src_ranks, self.src_rank = h, h["src_rank"], W.UNKNOWN_RANK
        self.stream_id, self.tag, self.byteorder = 0xFFFFFFFF, -1, W.LITTLEENDIAN
        if n:
            self.stream_id = int(h["stream_id"][0].item()) & 0xFFFFFFFF
            self.tag = int(h["tag"][0].item())
            bo = h["byteorder"]
            self.byteorder = int(bo[0].item()) if bool((bo == bo[0]).all()) else bo
            self._error = not bool(h["ok"].all())
        return self

    def to_message(self):
        """Message::send (src/Message.C:201-335): the bytes one write() puts on the link (uint8
        device tensor); a received wave goes out with the source ranks it came with."""
        cols = self.unpack(self.fmt.text)
        return self.codec.message_build(self.fmt, cols, self.n, stream_id=self.stream_id, tag=self.tag,
                                        src_rank=self.src_rank, src_ranks=self.src_ranks)

    # ---- accessors (include/mrnet/Packet.h:48-74) ----------------------------------
    def get_Tag(self):
        return self.tag

    def get_StreamId(self):
        return self.stream_id

    def get_FormatString(self):
        return self.fmt.text

    def get_InletNodeRank(self):
        return self.inlet_rank

    def get_SourceRank(self):
        return self.src_ranks if self.src_ranks is not None else self.src_rank

    def get_Buffer(self):
        return self.wire

    def get_BufferLen(self):
        return self.wire.lens if self.wire.lens is not None else self.fmt.fixed_bytes

    def get_Header(self):
        if self._hdrs is None:                                                   # encode_pdr_header
            self._hdrs, self._hdr_stride = self.codec.encode_headers(
                self.fmt, self.n, stream_id=self.stream_id, tag=self.tag, src_rank=self.src_rank,
                src_ranks=self.src_ranks)
        return self._hdrs

    def get_HeaderLen(self):
        return self.fmt.header_bytes

    def has_Error(self):
        return self._error

    def size(self):
        return self.n

    # ---- decode ------------------------------------------------------------------------
    def _decode(self):
        if not self._decoded:
            self.columns, self.ok = self.codec.decode(self.fmt, self.wire, self.byteorder)
            self._decoded = True
            if self.n and not bool(self.ok.all()):
                self._error = True

    def unpack(self, fmt):
        """Columns of the wave, or -1 when fmt is not the packets' format string."""
        self._decode()
        if fmt is None or fmt != self.fmt.text:
            return -1
        return self.columns

    def view(self, fmt):
        """The fields of ALL packets of the wave, in place: what unpack(fmt, ...) hands out one
        packet and one malloc'ed copy at a time (src/Packet.C:1065-1431), as typed views of the
        decoded device columns.  -1 when fmt is not the packets' format string (same check as
        unpack).  Per field: a tensor T[n] for a scalar; (counts[n], offs[n + 1], values) for an
        array or string field (element j of packet i is values[offs[i] + j]).  Nothing is copied
        and nothing synchronises: the tensors share the storage of the wave's columns (torch keeps
        that storage alive as long as a view exists; a C++ or C caller frees consumers first,
        columns last), and a consumer enqueues its work on the stream the decode ran on."""
        import ctypes as C
        from . import capi
        self._decode()
        views = (capi.FieldView * max(1, len(self.fmt)))()
        ok = capi.load().mrpdr_wave_view(self.fmt.handle, None if fmt is None else fmt.encode(),
                                         W._c_columns(self.columns), self.n, views)
        if not ok:
            return -1
        out = []
        for i, c in enumerate(self.columns):
            v = views[i]
            dt = W.torch_dtype(self.fmt, i)
            vals = None if c.values is None else c.values.view(torch.uint8).view(dt)
            assert vals is None or vals.data_ptr() == (v.values or 0)
            if v.kind == W.SCALAR:
                out.append(vals[: self.n])
            else:
                out.append((c.counts, c.offs, vals))
        return out

    def __getitem__(self, i):
        self._decode()
        return self.columns[i] if 0 <= i < len(self.columns) else None

    def values(self, i, dtype=None):
        """Host copy of field i (scalar: T[n]; array/string: (counts, values))."""
        c = self[i]
        dt = self.fmt.np_dtype(i) if dtype is None else dtype
        if self.fmt.kinds[i] == W.SCALAR:
            return W.Codec.to_numpy(c.values).view(dt)[: self.n]
        counts = W.Codec.to_numpy(c.counts).view(np.uint64)
        total = int(W.Codec.to_numpy(c.offs).view(np.uint64)[self.n])
        return counts, (W.Codec.to_numpy(c.values).view(dt)[:total] if total else np.zeros(0, dt))


_SCALAR_FORMATS = {"%c": W.DataType.CHAR_T, "%uc": W.DataType.UCHAR_T, "%hd": W.DataType.INT16_T,
                   "%uhd": W.DataType.UINT16_T, "%d": W.DataType.INT32_T, "%ud": W.DataType.UINT32_T,
                   "%ld": W.DataType.INT64_T, "%uld": W.DataType.UINT64_T, "%f": W.DataType.FLOAT_T,
                   "%lf": W.DataType.DOUBLE_T}


def group_by_format(hdrs, hdr_offsets, hdr_lens):
    """Host-side grouping of received packets by format string, in arrival order within a group.
    hdrs: the header bytes as they arrived (numpy uint8, host memory -- Message::recv leaves them
    there); hdr_offsets / hdr_lens: where each header lies.  The format string sits behind four
    32-bit words (src/Packet.C:557-575), its length in the header's own byte order (last byte,
    src/Packet.C:96).  Returns {format string: numpy array of packet indices}; packets whose
    header cannot hold a format string go under None.  The built-in filters only assert that a
    wave has one format (src/FilterDefinitions.C:104-105); a batched codec has to split first."""
    hdrs = np.ascontiguousarray(hdrs, dtype=np.uint8)
    groups = {}
    for i, (o, l) in enumerate(zip(np.asarray(hdr_offsets, dtype=np.int64), np.asarray(hdr_lens, dtype=np.int64))):
        key = None
        if l >= 16 + 1 + 8 + 1 and o >= 0 and o + l <= hdrs.size:
            h = hdrs[o:o + l]
            flen = int.from_bytes(h[12:16].tobytes(), "big" if h[l - 1] != W.LITTLEENDIAN else "little")
            if 1 <= flen <= l - 16 - 8 - 1 and h[16 + flen - 1] == 0:
                key = h[16:16 + flen - 1].tobytes().decode("latin-1")
        groups.setdefault(key, []).append(i)
    return {k: np.asarray(v, dtype=np.int64) for k, v in groups.items()}


def _run_filter(op, ipackets, shape):
    codec, fmt = ipackets.codec, ipackets.fmt
    if ipackets.n == 0 or ipackets.has_Error():
        return None
    if op == W.TFILTER_AVG:
        # "<T> %d" exactly, as tfilter_Avg compares the whole string (src/FilterDefinitions.C:525-544)
        parts = fmt.text.split(" ")
        if len(parts) != 2 or parts[1] != "%d" or parts[0] not in _SCALAR_FORMATS:
            return None
    elif fmt.text not in _SCALAR_FORMATS:
        return None             # the reference logs "invalid packet type" and emits nothing
    wire = ipackets.wire
    if wire.offsets is None:
        res, ok = codec.filter_wire(op, fmt, wire, ipackets.byteorder, shape)      # fused decode
    else:
        cols = ipackets.unpack(fmt.text)
        cnt = cols[1].values if op == W.TFILTER_AVG else None
        res, ok = codec.filter_columns(op, fmt.types[0], cols[0].values, cnt, shape)
    if int(ok.item()) == 0:
        return None             # e.g. an Avg whose counts sum to zero: nothing is emitted
    out = PacketWave.__new__(PacketWave)
    out.codec, out.fmt, out.n = codec, fmt, 1
    out.stream_id, out.tag = ipackets.get_StreamId(), ipackets.get_Tag()
    out.src_rank, out.src_ranks, out.inlet_rank = W.UNKNOWN_RANK, None, W.UNKNOWN_RANK
    out._hdrs, out._error, out.byteorder = None, False, W.LITTLEENDIAN
    # the 16 result bytes ARE the encoded data of the output packet
    out.wire = W.WireWave(res, 1, None, None, fmt.fixed_bytes)
    out.columns, out.ok, out._decoded = None, None, False
    out.result_ok = ok
    return out


def tfilter_Sum(ipackets, shape=W.SHAPE_FLAT):
    return _run_filter(W.TFILTER_SUM, ipackets, shape)


def tfilter_Max(ipackets, shape=W.SHAPE_FLAT):
    return _run_filter(W.TFILTER_MAX, ipackets, shape)


def tfilter_Min(ipackets, shape=W.SHAPE_FLAT):
    return _run_filter(W.TFILTER_MIN, ipackets, shape)


def tfilter_Avg(ipackets, shape=W.SHAPE_FLAT):
    return _run_filter(W.TFILTER_AVG, ipackets, shape)


def tfilter_IntSum(ipackets, shape=W.SHAPE_FLAT):
    """src/FilterDefinitions.C:71-88 (defined there, not registered): the int sum of '%d' packets."""
    if ipackets.fmt.text != "%d":
        return None
    return _run_filter(W.TFILTER_SUM, ipackets, shape)


def _from_columns(ipackets, fmt, columns):
    """One output packet with the stream id and tag of the first input, built from device columns."""
    return PacketWave(ipackets.codec, ipackets.get_StreamId(), ipackets.get_Tag(), fmt, columns, n=1)


def tfilter_ArrayConcat(ipackets):
    """One '%aT' / '%as' packet holding every input array in packet order.  Formats other than a
    single array token yield nothing, as in the reference (src/FilterDefinitions.C:764-768)."""
    fmt = ipackets.fmt
    if ipackets.n == 0 or ipackets.has_Error() or len(fmt) != 1 or fmt.kinds[0] not in (W.ARRAY, W.STRING_ARRAY):
        return None
    cols = ipackets.unpack(fmt.text)
    if ipackets.has_Error():
        return None
    return _from_columns(ipackets, fmt, [ipackets.codec.array_concat(cols[0], ipackets.n)])


_PERF_TYPES = {"%ad %ad %auld": W.DataType.UINT64_T, "%ad %ad %ald": W.DataType.INT64_T,
               "%ad %ad %alf": W.DataType.DOUBLE_T}


def tfilter_PerfData(ipackets, aggr_id, group=0):
    """Element-wise aggregation of performance-data packets (rank array, nelems array, data array).
    aggr_id: TFILTER_SUM / MIN / MAX / AVG, or TFILTER_ARRAY_CONCAT.  group > 0 declares a two-level tree (nodes of `group`
    children, then one node over their outputs)."""
    fmt, codec, n = ipackets.fmt, ipackets.codec, ipackets.n
    if n == 0 or ipackets.has_Error() or fmt.text not in _PERF_TYPES:
        return None
    cols = ipackets.unpack(fmt.text)
    if ipackets.has_Error():
        return None
    if aggr_id == W.TFILTER_ARRAY_CONCAT:
        # ranks, nelems and data of the packets that carry data, back to back
        mask = cols[2].counts
        return _from_columns(ipackets, fmt, [codec.array_concat_masked(cols[0], 4, mask, n),
                                             codec.array_concat_masked(cols[1], 4, mask, n),
                                             codec.array_concat_masked(cols[2], 8, mask, n)])
    # (group == 0, one invocation: for integer data the library folds large waves as a tree of
    # undivided nodes -- the same bits without a serial chain over all n packets; doubles are a chain)
    data, out_len, out_rank = codec.filter_perfdata(aggr_id, _PERF_TYPES[fmt.text], cols[0], cols[2], n, group)
    # output packet: ([-total_nranks], [len], data[0:len]); the length lives on the device
    one = torch.ones(1, dtype=torch.int64, device=codec.device)
    offs1 = torch.tensor([0, 1], dtype=torch.int64, device=codec.device)
    nelems = out_len.to(torch.int32)
    doffs = torch.cat([torch.zeros(1, dtype=torch.int64, device=codec.device), out_len])
    out_cols = [W.Column(out_rank, one, offs1, 1), W.Column(nelems, one, offs1, 1),
                W.Column(data, out_len, doffs, data.numel())]
    return _from_columns(ipackets, fmt, out_cols)


def tfilter_IntEqClass(ipackets):
    """src/FilterDefinitions.C:812-902: one '%aud %aud %aud' packet with the equivalence classes
    of all inputs merged by value."""
    fmt = ipackets.fmt
    if fmt.text != "%aud %aud %aud" or ipackets.n == 0 or ipackets.has_Error():
        return None
    cols = ipackets.unpack(fmt.text)
    if ipackets.has_Error():
        return None
    return _from_columns(ipackets, fmt, ipackets.codec.int_eq_class(cols[0], cols[1], cols[2], ipackets.n))


def aggr_FieldSums(ipackets, shape=W.SHAPE_FLAT):
    """The user filters of tests/test_DynamicFilters.C:12-58 (aggr_Count on '%d',
    aggr_CountOddsAndEvens on '%d %d') and Examples/IntegerAddition/IntegerAdditionFilter.C:15-35:
    unpack every packet, add each field up in its own C type, emit one packet of the same format.
    Any format of scalars only is taken; one fold per field, straight on the packed wire."""
    codec, fmt, n = ipackets.codec, ipackets.fmt, ipackets.n
    if n == 0 or ipackets.has_Error() or not fmt.is_fixed or len(fmt) == 0:
        return None
    out_bytes = torch.zeros(fmt.fixed_bytes, dtype=torch.uint8, device=codec.device)
    wire = ipackets.wire
    packed = wire.offsets is None and not isinstance(ipackets.byteorder, torch.Tensor)
    cols = None if packed else ipackets.unpack(fmt.text)
    if ipackets.has_Error():
        return None
    at = 0
    for f, w in enumerate(fmt.widths):
        if packed:
            res, _ = codec.filter_raw(W.TFILTER_SUM, fmt.types[f], wire.arena.data_ptr() + at, wire.stride, None, 0, n,
                                      int(ipackets.byteorder != W.LITTLEENDIAN), shape)
        else:
            res, _ = codec.filter_columns(W.TFILTER_SUM, fmt.types[f], cols[f].values, None, shape)
        out_bytes[at:at + w] = res[:w]
        at += w
    out = PacketWave.__new__(PacketWave)
    out.codec, out.fmt, out.n = codec, fmt, 1
    out.stream_id, out.tag = ipackets.get_StreamId(), ipackets.get_Tag()
    out.src_rank, out.src_ranks, out.inlet_rank = W.UNKNOWN_RANK, None, W.UNKNOWN_RANK
    out._hdrs, out._error, out.byteorder = None, False, W.LITTLEENDIAN
    out.wire = W.WireWave(out_bytes, 1, None, None, fmt.fixed_bytes)      # the sums ARE the encoded data
    out.columns, out.ok, out._decoded = None, None, False
    return out


aggr_Count = aggr_FieldSums               # tests/test_DynamicFilters.C:12-33
aggr_CountOddsAndEvens = aggr_FieldSums   # tests/test_DynamicFilters.C:35-58


def BitsetOr(ipackets, shape=W.SHAPE_FLAT):
    """Examples/FaultRecovery/FaultRecoveryFilter.C:107-128: one '%uld' packet, the OR of all inputs."""
    if ipackets.fmt.text != "%uld":
        return None
    return _run_filter(W.FILTER_OR, ipackets, shape)


def IntegerPercentiles(ipackets, state=None, shape=W.SHAPE_FLAT):
    """Examples/FaultRecovery/FaultRecoveryFilter.C:29-86 for waves of '%uld %ud %ud' packets
    (bit set, max, min): OR of the bit sets, maximum of the maxima, minimum of the minima, folded
    into the filter state the reference keeps in `void **filter_state`.  `state` is that state as
    a 16-byte device tensor in the packet's own layout (None = a fresh one: bits 0, max 0,
    min fr_range_max); returns (output PacketWave, new state).  Integer OR / max / min do not
    depend on the order, so the state simply is one more input of each fold."""
    codec, fmt, n = ipackets.codec, ipackets.fmt, ipackets.n
    if fmt.text != "%uld %ud %ud" or n == 0 or ipackets.has_Error() or ipackets.wire.offsets is not None \
            or isinstance(ipackets.byteorder, torch.Tensor):
        return None, state
    if state is None:
        state = torch.zeros(16, dtype=torch.uint8, device=codec.device)
        state[12:16] = torch.tensor(list((FR_RANGE_MAX).to_bytes(4, "little")), dtype=torch.uint8,
                                    device=codec.device)
    mask = torch.tensor(list(((1 << FR_BINS) - 1).to_bytes(8, "little")), dtype=torch.uint8, device=codec.device)
    # state and wave side by side: packet 0 is the state
    rows = torch.cat([state, ipackets.wire.arena[: n * 16]])
    base, swap = rows.data_ptr(), 0
    if ipackets.byteorder != W.LITTLEENDIAN:        # the state is native: fold the wave, then the state
        swap = 1
    out = torch.zeros(16, dtype=torch.uint8, device=codec.device)
    ok = torch.zeros(1, dtype=torch.int32, device=codec.device)
    for op, dt, off, w in ((W.FILTER_OR, W.DataType.UINT64_T, 0, 8), (W.TFILTER_MAX, W.DataType.UINT32_T, 8, 4),
                           (W.TFILTER_MIN, W.DataType.UINT32_T, 12, 4)):
        if swap:
            part, _ = codec.filter_raw(op, dt, base + 16 + off, 16, None, 0, n, 1, shape)
            pair = torch.cat([state[off:off + w], part[:w]])
            res, _ = codec.filter_raw(op, dt, pair.data_ptr(), w, None, 0, 2, 0, W.SHAPE_FLAT)
        else:
            res, _ = codec.filter_raw(op, dt, base + off, 16, None, 0, n + 1, 0, shape)
        if op == W.FILTER_OR:                       # fr_bin_set keeps fr_bins bits
            pair = torch.cat([res[:8], mask])
            res, _ = codec.filter_raw(W.FILTER_AND, dt, pair.data_ptr(), 8, None, 0, 2, 0, W.SHAPE_FLAT)
        out[off:off + w] = res[:w]
    opk = PacketWave.__new__(PacketWave)
    opk.codec, opk.fmt, opk.n = codec, fmt, 1
    opk.stream_id, opk.tag = ipackets.get_StreamId(), PROT_WAVE
    opk.src_rank, opk.src_ranks, opk.inlet_rank = W.UNKNOWN_RANK, None, W.UNKNOWN_RANK
    opk._hdrs, opk._error, opk.byteorder = None, False, W.LITTLEENDIAN
    opk.wire = W.WireWave(out, 1, None, None, 16)
    opk.columns, opk.ok, opk._decoded = None, None, False
    opk.result_ok = ok
    return opk, out.clone()


FR_RANGE_MAX, FR_BINS = 1000, 20      # Examples/FaultRecovery/FaultRecovery.h:12-13
PROT_WAVE = 102                       # FirstApplicationTag + 2 in that example's protocol enum (:17-19)


class Filter:
    """Filter::push_Packets for waves (src/Filter.C:24-131).  `func` follows the plugin calling
    convention of src/Filter.h:64-68 reduced to what a wave needs --
    func(ipackets, state, params) -> (list of output waves, new state) -- or is None, the NULL
    filter, which hands the input wave on untouched (src/Filter.C:51-58).  The filter state lives
    in the Filter between waves (void **filter_state there), parameters are a packet set with
    set_FilterParams (src/Filter.C:133-137), get_FilterState asks the optional state function for
    the state as a packet (src/Filter.C:119-131)."""

    def __init__(self, func=None, get_state=None):
        self._func, self._get_state = func, get_state
        self._state, self._params = None, None

    def set_FilterParams(self, params):
        self._params = params

    def push_Packets(self, ipackets):
        if self._func is None:
            return [ipackets]
        out, self._state = self._func(ipackets, self._state, self._params)
        return [p for p in out if p is not None]

    def get_FilterState(self, stream_id):
        if self._get_state is None:
            return None                                   # Packet::NullPacket
        return self._get_state(self._state, stream_id)


def _stateless(fn):
    return lambda ipackets, state, params: ([fn(ipackets)], state)


def _perfdata(ipackets, state, params):
    """tfilter_PerfData reads its aggregation id from the filter parameters, a '%d' packet
    (src/FilterDefinitions.C:925-935)."""
    if params is None:
        return [], state
    aggr = int(W.Codec.to_numpy(params.unpack("%d")[0].values).view(np.int32)[0])
    return [tfilter_PerfData(ipackets, aggr)], state


def _percentiles(ipackets, state, params):
    # kept between waves: (the 16 state bytes on the device, the codec that owns them)
    out, bits = IntegerPercentiles(ipackets, state[0] if state else None)
    return [out], (bits, ipackets.codec)


def _percentiles_state(state, stream_id):
    """IntegerPercentiles_get_state (Examples/FaultRecovery/FaultRecoveryFilter.C:88-105): the
    state as a '%uld %ud %ud' packet tagged PROT_WAVE on the asking stream."""
    if state is None or state[0] is None:
        return None
    pk = PacketWave.__new__(PacketWave)
    state, pk.codec = state
    pk.fmt, pk.n = W.Format("%uld %ud %ud"), 1
    pk.stream_id, pk.tag = stream_id, PROT_WAVE
    pk.src_rank, pk.src_ranks, pk.inlet_rank = W.UNKNOWN_RANK, None, W.UNKNOWN_RANK
    pk._hdrs, pk._error, pk.byteorder = None, False, W.LITTLEENDIAN
    pk.wire = W.WireWave(state.clone(), 1, None, None, 16)
    pk.columns, pk.ok, pk._decoded = None, None, False
    return pk


# the built-in transformation filters by name (ids are handed out at registration in the
# reference, src/FilterDefinitions.C:28-56, so names are what is stable)
BUILTIN_FILTERS = {
    "TFILTER_NULL": lambda: Filter(None),
    "TFILTER_SUM": lambda: Filter(_stateless(tfilter_Sum)),
    "TFILTER_AVG": lambda: Filter(_stateless(tfilter_Avg)),
    "TFILTER_MIN": lambda: Filter(_stateless(tfilter_Min)),
    "TFILTER_MAX": lambda: Filter(_stateless(tfilter_Max)),
    "TFILTER_ARRAY_CONCAT": lambda: Filter(_stateless(tfilter_ArrayConcat)),
    "TFILTER_INT_EQ_CLASS": lambda: Filter(_stateless(tfilter_IntEqClass)),
    "TFILTER_PERFDATA": lambda: Filter(_perfdata),
    "IntegerPercentiles": lambda: Filter(_percentiles, _percentiles_state),
    "BitsetOr": lambda: Filter(_stateless(BitsetOr)),
}
