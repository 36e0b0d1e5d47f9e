"""CPU-only checks of the drop-in boundary: the library loads, exports every symbol the
header declares, its host-side format compiler agrees with the oracle, and compute
entry points fail loudly without a GPU (no CPU fallback)."""
import ctypes as C
import os
import re

import pytest

from wavegen import O, ROOT

from mrnet_amd import capi


def header_symbols():
    text = open(os.path.join(ROOT, "include", "mrpdr.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mrpdr_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = capi.load()
    declared = header_symbols()
    assert declared, "no declarations found in include/mrpdr.h"
    for name in declared:
        assert hasattr(lib, name), "libmrpdr.so does not export %s" % name
    assert sorted(capi.SYMBOLS) == declared, "capi.SYMBOLS is out of step with include/mrpdr.h"
    assert lib.mrpdr_version().startswith(b"mrpdr")
    assert lib.mrpdr_local_byteorder() == 0


def test_no_torch_types_in_the_abi():
    text = open(os.path.join(ROOT, "include", "mrpdr.h")).read()
    assert 'extern "C"' in text
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)        # declarations only
    assert "torch" not in text and "at::" not in text and "std::" not in text


def test_header_is_plain_c():
    """The boundary is a C ABI: the header must compile as C99 (the lightweight back-end library
    of the reference is C, src/lightweight/)."""
    import subprocess
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", "-x", "c",
                        os.path.join(ROOT, "include", "mrpdr.h")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


FORMATS = ["%d", "%c %uc %hd %uhd %d %ud %ld %uld %f %lf %s", "%d %uld %lf %s %auc", "%af",
           "%lf %d", "%ad %ad %alf", "%Ac %Auc %Ahd %Auhd %Ad %Aud %Ald %Auld %Af %Alf",
           "%d\t%d\n%d%d", ""]


@pytest.mark.parametrize("fmt", FORMATS)
def test_format_compiler_matches_oracle(fmt):
    lib = capi.load()
    h = lib.mrpdr_fmt_compile(fmt.encode())
    assert h, capi.last_error()
    types = O.fmt_types(fmt)
    assert lib.mrpdr_fmt_nfields(h) == len(types)
    fixed = 0
    for i, t in enumerate(types):
        assert lib.mrpdr_fmt_field_type(h, i) == t
        assert lib.mrpdr_fmt_field_kind(h, i) == O.type_kind(t)
        assert lib.mrpdr_fmt_field_width(h, i) == O.type_width(t)
        fixed += {0: O.type_width(t), 1: 8, 2: 4}[O.type_kind(t)]
    assert lib.mrpdr_fmt_fixed_bytes(h) == fixed
    assert lib.mrpdr_fmt_is_fixed(h) == int(all(O.type_kind(t) == 0 for t in types))
    assert lib.mrpdr_fmt_string(h) == fmt.encode()
    # header length rule: 16 + strlen(fmt)+1 + 8 + 1 (src/Packet.C:557-593)
    assert lib.mrpdr_header_bytes(h) == len(O.header_encode(1, 2, 3, fmt))
    lib.mrpdr_fmt_free(h)


@pytest.mark.parametrize("fmt", ["%q", "%lld", "%d %x", "%" + "d %" * 40 + "d"])
def test_format_compiler_rejects(fmt):
    lib = capi.load()
    assert not lib.mrpdr_fmt_compile(fmt.encode())
    assert capi.last_error()


def test_format_token_bound_is_exact():
    """32 tokens compile, 33 are refused with a message (the reference's tokenizer has no bound,
    src/Packet.C:608-619; the field table of this library is a kernel argument of fixed size)."""
    lib = capi.load()
    h = lib.mrpdr_fmt_compile(" ".join(["%d"] * 32).encode())
    assert h and lib.mrpdr_fmt_nfields(h) == 32
    lib.mrpdr_fmt_free(h)
    assert not lib.mrpdr_fmt_compile(" ".join(["%d"] * 33).encode())
    assert "32" in capi.last_error()


def test_compute_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    lib = capi.load()
    assert not lib.mrpdr_ctx_create(0, 1024)
    assert "HIP device" in capi.last_error()
    from mrnet_amd.wave import Codec, PdrError
    with pytest.raises(PdrError):
        Codec()


def test_product_does_not_import_the_oracle():
    pkg = os.path.join(ROOT, "mrnet_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".hip", ".h", ".hpp", ".cc", ".cpp")):
                text = open(os.path.join(dirpath, fn), errors="ignore").read()
                assert "pdr_oracle" not in text and "oracle." not in text and "import oracle" not in text, \
                    "%s refers to the oracle" % fn


def test_group_by_format_host_logic():
    """Packets of several formats and both byte orders split by format string (host logic only)."""
    import numpy as np
    from mrnet_amd.packet import group_by_format
    fmts = ["%d", "%d %uld %lf %s %auc", "%d", "%af", "%d %uld %lf %s %auc", "%d"]
    hdrs, offs, lens = [], [], []
    at = 3                                                   # headers need not start at offset 0
    blob = bytearray(b"\xee" * at)
    for i, f in enumerate(fmts):
        h = O.header_encode(7 + i, 100, i, f).copy()
        if i % 2:                                            # as a big-endian sender writes it
            for a in (0, 4, 8, 12):
                h[a:a + 4] = h[a:a + 4][::-1].copy()
            e = 16 + len(f) + 1
            h[e:e + 8] = h[e:e + 8][::-1].copy()
            h[-1] = 1
            assert O.header_decode(h)["fmt"] == f
        offs.append(len(blob)); lens.append(h.size)
        blob += h.tobytes() + b"\xee\xee"
    offs.append(len(blob)); lens.append(5)                   # a runt
    blob += b"\x00" * 5
    g = group_by_format(np.frombuffer(bytes(blob), dtype=np.uint8), offs, lens)
    assert sorted(k for k in g if k is not None) == sorted(set(fmts))
    assert g["%d"].tolist() == [0, 2, 5] and g["%af"].tolist() == [3]
    assert g["%d %uld %lf %s %auc"].tolist() == [1, 4] and g[None].tolist() == [6]


def test_wave_view_is_host_logic_with_unpack_format_check():
    """mrpdr_wave_view touches no device: it maps columns to per-field views under the format
    check of Packet::unpack (src/Packet.C:527-532) -- runs without a GPU."""
    lib = capi.load()
    fmt = b"%d %af %s %as"
    h = lib.mrpdr_fmt_compile(fmt)
    cols = (capi.Column * 4)()
    for i in range(4):
        cols[i] = capi.Column(0x1000 * (i + 1), 0x100 * (i + 1), 0x10 * (i + 1), 5, None, None, 0)
    cols[0] = capi.Column(0x1000, None, None, 7, None, None, 0)
    cols[3] = capi.Column(0x4000, 0x400, 0x40, 5, 0x44, 0x48, 3)
    views = (capi.FieldView * 4)()
    assert lib.mrpdr_wave_view(h, b"%d %af %s", cols, 7, views) == 0 and "format" in capi.last_error()
    assert lib.mrpdr_wave_view(h, None, cols, 7, views) == 0
    assert lib.mrpdr_wave_view(h, fmt, cols, 7, views) == 1
    assert [v.kind for v in views] == [0, 1, 2, 3] and [v.width for v in views] == [4, 4, 1, 1]
    assert views[0].values == 0x1000 and views[0].counts is None and views[0].offs is None and views[0].n == 7
    assert views[1].values == 0x2000 and views[1].counts == 0x200 and views[1].offs == 0x20
    assert views[3].sub_counts == 0x44 and views[3].sub_offs == 0x48 and views[2].sub_offs is None
    assert views[1].type == O.fmt_types("%af")[0]
    bad = (capi.Column * 4)(*[capi.Column(None, None, None, 0, None, None, 0)] * 4)
    assert lib.mrpdr_wave_view(h, fmt, bad, 7, views) == 0          # columns that hold nothing
    lib.mrpdr_fmt_free(h)
