"""CPU-only: the C-ABI library builds for gfx950, loads, and exports every function include/*.h declares.
No compute call is made (there is no GPU here); the product has no CPU path and must say so."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions(header):
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    txt = re.sub(r"//[^\n]*", "", txt)
    txt = re.sub(r"typedef[^;]*\(\s*\*\s*\w+\s*\)\s*\([^;]*\)\s*;", "", txt)  # function-pointer typedefs
    names = re.findall(r"\b([A-Za-z_]\w*)\s*\([^;{]*\)\s*;", txt)
    return sorted(set(n for n in names if n not in ("defined",)))


@pytest.fixture(scope="module")
def hiplib():
    import speedy_amd
    speedy_amd.build()
    return speedy_amd.lib()


@pytest.mark.parametrize("header", ["speedy_hip.h", "sonic2.h", "speedy.h"])
def test_every_declared_symbol_is_exported(hiplib, header):
    names = declared_functions(header)
    assert len(names) >= 15
    raw = ctypes.CDLL(os.path.join(ROOT, "speedy_amd", "lib", "libspeedy_hip.so"))
    missing = [n for n in names if not hasattr(raw, n)]
    assert not missing, missing


def test_python_binding_covers_the_headers():
    from speedy_amd._lib import SYMBOLS
    declared = set(declared_functions("speedy_hip.h")) | set(declared_functions("sonic2.h")) | set(declared_functions("speedy.h"))
    assert declared <= set(SYMBOLS), sorted(declared - set(SYMBOLS))


def test_reference_api_surface_present():
    """Every function of the reference's public header (sonic2.h:54-125) exists with the same name."""
    ref = ["sonicCreateStream", "sonicDestroyStream", "sonicWriteShortToStream", "sonicReadShortFromStream",
           "sonicWriteFloatToStream", "sonicReadFloatFromStream", "sonicSetRate", "sonicSetSpeed", "sonicFlushStream",
           "sonicEnableNonlinearSpeedup", "sonicSetDurationFeedbackStrength", "getSonicBufferSize",
           "sonicTensionCallback", "getSonicTensionCallback", "sonicSpeedCallback", "getSonicSpeedCallback",
           "sonicFeaturesCallback", "getSonicFeaturesCallback", "sonicSpectrogramCallback",
           "getSonicSpectrogramCallback", "sonicNormalizedSpectrogramCallback",
           "getSonicNormalizedSpectrogramCallback", "sonicSpectrogramSize"]
    assert set(ref) <= set(declared_functions("sonic2.h"))


def test_no_cpu_fallback(hiplib):
    """Without a GPU the product refuses to run instead of silently computing on the host."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from speedy_amd.batch import Plan
    with pytest.raises(RuntimeError):
        Plan(16000)
    assert not hiplib.sonicCreateStream(16000, 1)
    assert b"no HIP device" in hiplib.speedyHipLastError() or b"plan" in hiplib.speedyHipLastError()


def test_product_does_not_import_the_oracle():
    """The oracle is test infrastructure: nothing under speedy_amd/ may reference it."""
    bad = []
    for dirpath, _, files in os.walk(os.path.join(ROOT, "speedy_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                t = open(os.path.join(dirpath, f), errors="ignore").read()
                if re.search(r"^\s*(from|import)\s+oracle|#include\s+\"[^\"]*orc_|liborc", t, flags=re.M):
                    bad.append(f)
    assert not bad, bad


@pytest.mark.parametrize("header", ["speedy_hip.h", "sonic2.h", "speedy.h"])
def test_headers_are_plain_c(header, tmp_path):
    """The boundary is a C ABI: every header compiles as strict C99 (and as C++) on its own, with no HIP header in reach."""
    import subprocess
    src = tmp_path / "t.c"
    src.write_text('#include "%s"\nint main(void) { return 0; }\n' % header)
    inc = os.path.join(ROOT, "include")
    subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-fsyntax-only", "-I", inc, str(src)])
    subprocess.check_call(["g++", "-std=c++11", "-pedantic", "-Wall", "-Wextra", "-Werror", "-fsyntax-only", "-x", "c++",
                           "-I", inc, str(src)])


def test_c_example_builds():
    """tools/batch_example.c (INTEGRATION.md section 2 as a program) builds with gcc -std=c99 -pedantic -Werror."""
    import subprocess
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "speedy_amd", "csrc"), "example"])
    assert os.path.exists(os.path.join(ROOT, "speedy_amd", "lib", "batch_example"))


@pytest.mark.parametrize("header", ["sonic.h", "wave.h"])
def test_compat_headers_are_plain_c_and_exported(hiplib, header, tmp_path):
    """include/compat/: what the reference's callers include from libsonic (speedy_wave.cc:24,27; sonic_test.cc:37)."""
    import subprocess
    inc = os.path.join(ROOT, "include")
    for pre in ("", "#define SONIC_INTERNAL 1\n"):
        src = tmp_path / "t.c"
        src.write_text(pre + '#include "%s"\n#include "sonic2.h"\nint main(void) { return 0; }\n' % header)
        subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-fsyntax-only",
                               "-I", os.path.join(inc, "compat"), "-I", inc, str(src)])
    txt = open(os.path.join(inc, "compat", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    renames = dict(re.findall(r"#define\s+(sonic\w+)\s+(sonicInt\w+)", txt))
    names = re.findall(r"\b([A-Za-z_]\w*)\s*\([^;{]*\)\s*;", txt)
    raw = ctypes.CDLL(os.path.join(ROOT, "speedy_amd", "lib", "libspeedy_hip.so"))
    missing = [n for n in names if not hasattr(raw, n)] + [v for v in renames.values() if not hasattr(raw, v)]
    assert len(names) >= 5 and not missing, missing


REFERENCE = "/root/reference"


@pytest.mark.skipif(not os.path.exists(os.path.join(REFERENCE, "speedy_wave.cc")), reason="build container only: /root/reference absent")
def test_reference_cli_source_compiles_and_links_unchanged(hiplib, tmp_path):
    """INTEGRATION.md section 1, literally: the reference's own CLI source -- compiled where it lies, never copied --
    builds against include/compat (libsonic's two headers) + the reference's own sonic2.h / speedy.h and links
    libspeedy_hip.so with no other library.  (Running it needs a GPU: tests/test_gpu_cli.py.)"""
    import subprocess
    exe = tmp_path / "speedy_wave_ref"
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-w", "-I", os.path.join(ROOT, "include", "compat"), "-I", REFERENCE,
                           os.path.join(REFERENCE, "speedy_wave.cc"), "-L", os.path.join(ROOT, "speedy_amd", "lib"),
                           "-lspeedy_hip", "-Wl,-rpath," + os.path.join(ROOT, "speedy_amd", "lib"), "-Wl,-rpath,/opt/rocm/lib",
                           "-o", str(exe)])
    assert exe.exists()
    # ... and against this repo's headers alone (include/ in front of the reference's directory cannot be forced for
    # quoted includes, so the reference's two headers are masked by compiling a two-line wrapper from include/)
    wrap = tmp_path / "wrap.cc"
    wrap.write_text('#include "sonic.h"\nextern "C" {\n#include "wave.h"\n#include "sonic2.h"\n#include "speedy.h"\n}\n'
                    'int main() { sonicStream s = sonicCreateStream(16000, 1); if (s) sonicDestroyStream(s);\n'
                    '  return kTemporalHysteresisFuture == 12 ? 0 : 1; }\n')
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include", "compat"),
                           "-I", os.path.join(ROOT, "include"), str(wrap), "-L", os.path.join(ROOT, "speedy_amd", "lib"),
                           "-lspeedy_hip", "-Wl,-rpath,/opt/rocm/lib", "-o", str(tmp_path / "wrap")])
