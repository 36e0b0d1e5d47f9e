"""ctypes declarations of the C ABI in include/mrpdr.h (mrnet_amd/lib/libmrpdr.so).

There is no CPU fallback: if the HIP library is missing, importing this module
raises, and every compute entry point fails without a GPU.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# MRPDR_LIB names another build of the same library (kernel experiments); default: in-tree
LIB_PATH = os.environ.get("MRPDR_LIB") or os.path.join(_HERE, "lib", "libmrpdr.so")

# every symbol include/mrpdr.h declares
SYMBOLS = (
    "mrpdr_version", "mrpdr_last_error", "mrpdr_local_byteorder",
    "mrpdr_ctx_create", "mrpdr_ctx_destroy", "mrpdr_ctx_status",
    "mrpdr_ctx_perf_enable", "mrpdr_ctx_perf_collect", "mrpdr_ctx_perf_plain_copy",
    "mrpdr_fmt_compile", "mrpdr_fmt_free", "mrpdr_fmt_nfields", "mrpdr_fmt_field_type",
    "mrpdr_fmt_field_kind", "mrpdr_fmt_field_width", "mrpdr_fmt_is_fixed",
    "mrpdr_fmt_fixed_bytes", "mrpdr_fmt_payload_phase", "mrpdr_fmt_string",
    "mrpdr_wave_sizeof", "mrpdr_wave_encode", "mrpdr_wave_encode_begin", "mrpdr_wave_encode_finish",
    "mrpdr_wave_decode_begin", "mrpdr_wave_decode_finish", "mrpdr_wave_decode",
    "mrpdr_wave_decode_mixed_begin", "mrpdr_wave_decode_mixed_finish", "mrpdr_wave_decode_mixed", "mrpdr_wave_view",
    "mrpdr_header_bytes", "mrpdr_header_bytes_dest", "mrpdr_wave_encode_headers", "mrpdr_wave_decode_headers",
    "mrpdr_message_preamble_bytes", "mrpdr_message_write_preamble", "mrpdr_message_parse_sizes",
    "mrpdr_message_write_frame", "mrpdr_message_parse",
    "mrpdr_wave_filter", "mrpdr_filter_is_chained", "mrpdr_wave_filter_partial", "mrpdr_wave_filter_combine",
    "mrpdr_wave_decode_filter", "mrpdr_wave_filter_mixed", "mrpdr_wave_filter_partial_mixed", "mrpdr_wave_filter_noalloc", "mrpdr_wave_filter_stats", "mrpdr_wave_array_concat", "mrpdr_wave_array_concat_masked", "mrpdr_perfdata_scratch_bytes", "mrpdr_wave_perfdata",
    "mrpdr_perfdata_levels_scratch_bytes", "mrpdr_wave_perfdata_levels", "mrpdr_perfdata_state_bytes",
    "mrpdr_perfdata_is_chained", "mrpdr_wave_perfdata_partial", "mrpdr_wave_perfdata_combine",
    "mrpdr_eqclass_scratch_bytes", "mrpdr_wave_int_eq_class",
)


class Column(C.Structure):
    """struct mrpdr_column"""
    _fields_ = [("values", C.c_void_p), ("counts", C.c_void_p), ("offs", C.c_void_p),
                ("capacity", C.c_uint64), ("sub_counts", C.c_void_p), ("sub_offs", C.c_void_p),
                ("sub_capacity", C.c_uint64)]


class FieldView(C.Structure):
    """struct mrpdr_field_view"""
    _fields_ = [("values", C.c_void_p), ("counts", C.c_void_p), ("offs", C.c_void_p),
                ("sub_counts", C.c_void_p), ("sub_offs", C.c_void_p), ("n", C.c_uint64),
                ("type", C.c_int32), ("kind", C.c_int32), ("width", C.c_int32), ("reserved", C.c_int32)]


class Wire(C.Structure):
    """struct mrpdr_wire"""
    _fields_ = [("base", C.c_void_p), ("offsets", C.c_void_p), ("lens", C.c_void_p),
                ("stride", C.c_uint64), ("bytes", C.c_uint64)]


_lib = None


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "mrnet_amd: %s is missing -- build it with `python -c 'import __graft_entry__ as g; "
            "g.build()'` (hipcc --offload-arch=gfx950); there is no CPU fallback" % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    vp, u64, i32, u32 = C.c_void_p, C.c_uint64, C.c_int32, C.c_uint32
    sig = {
        "mrpdr_version": (C.c_char_p, []),
        "mrpdr_last_error": (C.c_char_p, []),
        "mrpdr_local_byteorder": (i32, []),
        "mrpdr_ctx_create": (vp, [i32, u64]),
        "mrpdr_ctx_destroy": (None, [vp]),
        "mrpdr_ctx_status": (i32, [vp, vp, C.POINTER(u32)]),
        "mrpdr_ctx_perf_enable": (i32, [vp, i32]),
        "mrpdr_ctx_perf_collect": (i32, [vp, vp, C.POINTER(C.c_double), C.POINTER(u64)]),
        "mrpdr_ctx_perf_plain_copy": (i32, [vp, vp, vp, u64, i32, C.POINTER(C.c_double), vp]),
        "mrpdr_fmt_compile": (vp, [C.c_char_p]),
        "mrpdr_fmt_free": (None, [vp]),
        "mrpdr_fmt_nfields": (i32, [vp]),
        "mrpdr_fmt_field_type": (i32, [vp, i32]),
        "mrpdr_fmt_field_kind": (i32, [vp, i32]),
        "mrpdr_fmt_field_width": (i32, [vp, i32]),
        "mrpdr_fmt_is_fixed": (i32, [vp]),
        "mrpdr_fmt_fixed_bytes": (u64, [vp]),
        "mrpdr_fmt_payload_phase": (u32, [vp, u32]),
        "mrpdr_fmt_string": (C.c_char_p, [vp]),
        "mrpdr_wave_sizeof": (i32, [vp, vp, C.POINTER(Column), u64, u32, u32, vp, vp, C.POINTER(u64), vp]),
        "mrpdr_wave_encode": (i32, [vp, vp, C.POINTER(Column), u64, C.POINTER(Wire), vp]),
        "mrpdr_wave_encode_begin": (i32, [vp, vp, C.POINTER(Column), u64, C.POINTER(Wire), vp]),
        "mrpdr_wave_encode_finish": (i32, [vp, vp, C.POINTER(Column), u64, C.POINTER(Wire), vp]),
        "mrpdr_wave_decode_begin": (i32, [vp, vp, C.POINTER(Wire), u64, i32, C.POINTER(Column), vp,
                                          C.POINTER(u64), vp]),
        "mrpdr_wave_decode_finish": (i32, [vp, vp, C.POINTER(Wire), u64, i32, C.POINTER(Column), vp]),
        "mrpdr_wave_decode": (i32, [vp, vp, C.POINTER(Wire), u64, i32, C.POINTER(Column), vp, vp]),
        "mrpdr_wave_decode_mixed_begin": (i32, [vp, vp, C.POINTER(Wire), u64, vp, C.POINTER(Column), vp,
                                                C.POINTER(u64), vp]),
        "mrpdr_wave_decode_mixed_finish": (i32, [vp, vp, C.POINTER(Wire), u64, vp, C.POINTER(Column), vp]),
        "mrpdr_wave_decode_mixed": (i32, [vp, vp, C.POINTER(Wire), u64, vp, C.POINTER(Column), vp, vp]),
        "mrpdr_wave_view": (i32, [vp, C.c_char_p, C.POINTER(Column), u64, C.POINTER(FieldView)]),
        "mrpdr_header_bytes": (u64, [vp]),
        "mrpdr_header_bytes_dest": (u64, [vp, u64]),
        "mrpdr_wave_encode_headers": (i32, [vp, vp, u64, u32, vp, i32, vp, u32, vp, vp, u64, vp, vp, u64, vp]),
        "mrpdr_message_preamble_bytes": (u64, [u64]),
        "mrpdr_message_write_preamble": (i32, [vp, u64, u64, vp, vp, vp]),
        "mrpdr_message_parse_sizes": (i32, [vp, vp, u64, vp, vp, vp, vp, C.POINTER(u64), vp]),
        "mrpdr_message_write_frame": (i32, [vp, vp, u64, vp, vp, u32, vp, i32, vp, u32, vp, vp, vp]),
        "mrpdr_message_parse": (i32, [vp, vp, vp, u64, u64, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, u64, vp, vp]),
        "mrpdr_wave_decode_headers": (i32, [vp, vp, u64, vp, u64, vp, vp, u64, vp, vp, vp, vp, vp, vp, u64, vp, vp]),
        "mrpdr_wave_filter": (i32, [vp, i32, i32, i32, vp, u64, vp, u64, u64, i32, vp, vp, vp]),
        "mrpdr_filter_is_chained": (i32, [i32, i32]),
        "mrpdr_wave_filter_partial": (i32, [vp, i32, i32, vp, u64, vp, u64, u64, i32, vp, vp, vp]),
        "mrpdr_wave_filter_partial_mixed": (i32, [vp, i32, i32, vp, u64, vp, u64, u64, vp, vp, vp, vp]),
        "mrpdr_wave_filter_combine": (i32, [vp, i32, i32, vp, u64, vp, vp, vp]),
        "mrpdr_wave_filter_mixed": (i32, [vp, i32, i32, i32, vp, u64, vp, u64, u64, vp, vp, vp, vp]),
        "mrpdr_wave_decode_filter": (i32, [vp, vp, C.POINTER(Wire), u64, i32, C.POINTER(Column), vp, i32, i32, vp, vp, vp]),
        "mrpdr_wave_filter_noalloc": (i32, [i32, i32, i32, vp, u64, vp, u64, u64, i32, vp, vp, vp]),
        "mrpdr_wave_filter_stats": (i32, [vp, i32, vp, u64, vp, u64, u64, i32, vp, vp, vp]),
        "mrpdr_wave_array_concat": (i32, [vp, C.POINTER(Column), u64, C.POINTER(Column), vp]),
        "mrpdr_wave_array_concat_masked": (i32, [vp, C.POINTER(Column), u32, vp, u64, C.POINTER(Column), vp]),
        "mrpdr_perfdata_scratch_bytes": (u64, [u64, u64, u64]),
        "mrpdr_perfdata_levels_scratch_bytes": (u64, [u64, C.POINTER(u64), i32, u64]),
        "mrpdr_wave_perfdata_levels": (i32, [vp, i32, i32, C.POINTER(Column), C.POINTER(Column), u64, C.POINTER(u64), i32,
                                             vp, u64, vp, vp, vp, u64, vp]),
        "mrpdr_perfdata_state_bytes": (u64, [u64]),
        "mrpdr_perfdata_is_chained": (i32, [i32, i32]),
        "mrpdr_wave_perfdata_partial": (i32, [vp, i32, i32, C.POINTER(Column), C.POINTER(Column), u64, u64, vp, vp, vp, u64, vp]),
        "mrpdr_wave_perfdata_combine": (i32, [vp, i32, i32, vp, u64, u64, vp, u64, vp, vp, vp]),
        "mrpdr_eqclass_scratch_bytes": (u64, [u64]),
        "mrpdr_wave_int_eq_class": (i32, [vp, C.POINTER(Column), C.POINTER(Column), C.POINTER(Column), u64,
                                          C.POINTER(Column), C.POINTER(Column), C.POINTER(Column), vp, u64, vp]),
        "mrpdr_wave_perfdata": (i32, [vp, i32, i32, C.POINTER(Column), C.POINTER(Column), u64, u64, vp, u64,
                                      vp, vp, vp, u64, vp]),
    }
    for name in SYMBOLS:
        fn = getattr(L, name)           # AttributeError if the library lacks a declared symbol
        fn.restype, fn.argtypes = sig[name]
    _lib = L
    return L


def last_error():
    return load().mrpdr_last_error().decode()
