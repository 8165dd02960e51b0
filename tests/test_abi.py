"""The C-ABI library loads without a GPU, exports every symbol include/glove_hip.h declares and
its structs have the layout the ctypes binding assumes.  No compute calls here."""
import ctypes as C
import re
import subprocess
from pathlib import Path

import pytest

REPO = Path(__file__).resolve().parent.parent
HEADER = REPO / "include" / "glove_hip.h"


@pytest.fixture(scope="module")
def lib():
    from trainer import hip_api
    if not hip_api.LIB_PATH.exists():
        import __graft_entry__
        __graft_entry__.build()
    return hip_api.load_library()


def declared_functions():
    text = re.sub(r"/\*.*?\*/", "", HEADER.read_text(), flags=re.S)
    return sorted(set(re.findall(r"\b(glove_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported(lib):
    from trainer import hip_api
    names = declared_functions()
    assert len(names) >= 16
    assert set(names) == set(hip_api.EXPORTED_SYMBOLS)
    for n in names:
        assert hasattr(lib, n), n
    assert lib.glove_abi_version() == hip_api.GLOVE_ABI_VERSION


def test_size_queries_are_pure_host_functions(lib):
    assert lib.glove_dense_grad_layout(100, 100, 64, None) == 2 * 100 * 64 + 2 * 100 + 8       # V % 4 == 0: no padding
    assert lib.glove_dense_grad_layout(101, 101, 8, None) == 101 * 8 + 104 + 101 * 8 + 104 + 8   # sections 16-B aligned
    small, big = lib.glove_step_workspace_bytes(1024, 1024, 64), lib.glove_step_workspace_bytes(4096, 4096, 64)
    assert 0 < small < big
    assert lib.glove_plan_workspace_bytes(1024, 1000) > 1024 * 4 * 5
    assert lib.glove_topk_workspace_bytes(8, 1000, 20) >= 8 * 1000 * 4


def test_struct_layout_matches_the_c_header(tmp_path):
    from trainer import hip_api
    src = tmp_path / "layout.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "glove_hip.h"\nint main(void){\n'
                   'printf("%zu %zu %zu %zu %zu %zu %zu %zu ", sizeof(glove_tables), sizeof(glove_hyper), '
                   'sizeof(glove_plan), offsetof(glove_tables, scalars), offsetof(glove_hyper, inv_batch), '
                   'offsetof(glove_plan, host_counts), offsetof(glove_plan, r_to_c), offsetof(glove_plan, c_crec));\n'
                   'printf("%zu %zu %zu ", offsetof(glove_tables, R), offsetof(glove_hyper, sides), offsetof(glove_tables, d_model));\n'
                   'printf("%zu %zu %zu %zu %zu ", offsetof(glove_tables, R_ver), offsetof(glove_hyper, step_form), '
                   'offsetof(glove_plan, V_row), sizeof(glove_packed_list), offsetof(glove_packed_list, n));\n'
                   'printf("%zu %zu\\n", (size_t)GLOVE_FUSED_STEP_BYTES, (size_t)GLOVE_PACKED_ENTRY_FLOATS(300));\n'
                   'return 0;}\n')
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-I", str(REPO / "include"), str(src), "-o", str(exe)], check=True)
    got = [int(x) for x in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()]
    T, H, P = hip_api.GloveTables, hip_api.GloveHyper, hip_api.GlovePlan
    assert got == [C.sizeof(T), C.sizeof(H), C.sizeof(P), T.scalars.offset, H.inv_batch.offset,
                   P.host_counts.offset, P.r_to_c.offset, P.c_crec.offset, T.R.offset, H.sides.offset, T.d_model.offset,
                   T.R_ver.offset, H.step_form.offset, P.V_row.offset, C.sizeof(hip_api.GlovePackedList),
                   hip_api.GlovePackedList.n.offset, hip_api.FUSED_STEP_BYTES, 300 + 4]


def test_missing_library_is_an_error_not_a_fallback(tmp_path):
    from trainer import hip_api
    with pytest.raises(hip_api.GloveHipError, match="no CPU fallback"):
        hip_api.load_library(tmp_path / "libglove_hip.so")


def test_every_export_has_a_caller():
    """No dead surface: every function include/glove_hip.h declares is reached from the product (trainer/, bench.py,
    __graft_entry__.py) — directly, or through the GloveHip method that wraps it — and from at least one test."""
    import glob
    from trainer import hip_api
    src = (REPO / "glove-tensorflow_amd" / "trainer" / "hip_api.py").read_text()
    wrappers = {}                                   # symbol -> names of the hip_api functions / methods whose body calls it
    for m in re.finditer(r"^( *)def (\w+)\(.*?(?=^\1def |^class |\Z)", src, re.S | re.M):
        for sym in set(re.findall(r"(?:lib|load_library\(\))\.(glove_\w+)", m.group(0))) | set(re.findall(r"\(\"(glove_\w+)\", plan", m.group(0))):
            wrappers.setdefault(sym, set()).add(m.group(2))
    product = {f: open(f).read() for f in glob.glob(str(REPO / "glove-tensorflow_amd" / "trainer" / "*.py")) if not f.endswith("hip_api.py")}
    for f in ("bench.py", "__graft_entry__.py"):
        product[f] = (REPO / f).read_text()
    tests = {f: open(f).read() for f in glob.glob(str(REPO / "tests" / "*.py"))}

    def reached(sym, files, seen=()):
        names = {sym} | wrappers.get(sym, set())
        if any(re.search(r"\b%s\b" % re.escape(n), t) for n in names for t in files.values()):
            return True
        # a wrapper that only other hip_api functions call (Plan.compact inside build_plan, workspace queries inside steps)
        for n in wrappers.get(sym, set()):
            for m in re.finditer(r"^( *)def (\w+)\(.*?(?=^\1def |^class |\Z)", src, re.S | re.M):
                outer = m.group(2)
                if outer != n and outer not in seen and re.search(r"\b%s\(" % re.escape(n), m.group(0)):
                    if any(re.search(r"\b%s\b" % re.escape(outer), t) for t in files.values()):
                        return True
        return False
    names = declared_functions()
    assert len(names) <= 42, "the C ABI stays thin: %d exports" % len(names)
    dead = [n for n in names if n != "glove_abi_version" and not reached(n, product)]
    untested = [n for n in names if not reached(n, tests)]
    assert not dead, "exports without a caller in the product: %s" % dead
    assert not untested, "exports no test reaches: %s" % untested
