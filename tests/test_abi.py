"""CPU: the C-ABI library loads and exports every symbol that include/*.h declares
(no compute calls -- there is no GPU here), and the drop-in shim object exports exactly
the reference's eight public symbols (src/smatrix.h:87-94)."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(ROOT, "libsmatrix_amd", "lib")


@pytest.fixture(scope="module")
def built():
    if not os.path.exists(os.path.join(LIBDIR, "smatrix.so")):
        subprocess.run(["make", "-C", os.path.join(ROOT, "libsmatrix_amd", "csrc")], check=True)
    return LIBDIR


def declared(header):
    src = open(os.path.join(ROOT, "include", header)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b((?:smatrix|smx)_[a-z0-9_]+)\s*\(", src)))


def exported(path):
    out = subprocess.run(["nm", "-D", "--defined-only", path], check=True, capture_output=True, text=True).stdout
    return {ln.split()[-1] for ln in out.splitlines() if " T " in ln}


def test_every_declared_symbol_is_exported(built):
    syms = exported(os.path.join(built, "smatrix.so"))
    for h in ("smatrix.h", "smatrix_batch.h", "smatrix_shard.h", "smx_probe.h", "smx_stream.h"):
        names = declared(h)
        assert names, h
        missing = [n for n in names if n not in syms]
        assert not missing, (h, missing)


def test_ctypes_binding_loads(built):
    from libsmatrix_amd import _lib
    lib = _lib.load()
    assert lib.smx_fmix32(0) == 0 and lib.smx_fmix32(1) == 0x514E28B7   # murmur3 fmix32(1)


def test_shim_object_exports_the_reference_abi(built):
    out = subprocess.run(["nm", "--defined-only", os.path.join(built, "smatrix.o")], check=True,
                         capture_output=True, text=True).stdout
    syms = sorted(ln.split()[-1] for ln in out.splitlines() if " T " in ln)
    assert syms == sorted("smatrix_" + n for n in
                          ("open", "close", "get", "set", "incr", "decr", "rowlen", "getrow"))


def test_open_without_gpu_fails_loudly(built):
    """no CPU fallback: on a box without a HIP device smatrix_open returns NULL"""
    import libsmatrix_amd
    if libsmatrix_amd.device_available():
        pytest.skip("a GPU is present")
    with pytest.raises(ValueError):
        libsmatrix_amd.SparseMatrix()


def test_headers_and_c_caller_compile_as_c99(built, tmp_path):
    """include/*.h are C headers for C callers (JNI, Ruby, plain C): every one of them, and the C test program that uses
    the drop-in calls, the batch API, flush and the shard helpers, compiles as strict C99 and links against smatrix.so"""
    inc = os.path.join(ROOT, "include")
    for h in sorted(os.listdir(inc)):
        src = tmp_path / ("inc_" + h + ".c")
        src.write_text('#include "%s"\nint main(void) { return 0; }\n' % h)
        subprocess.run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-I" + inc, "-c", str(src), "-o", str(tmp_path / "o.o")], check=True)
    exe = str(tmp_path / "abi_known_answers")
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I" + inc, os.path.join(ROOT, "tests", "c", "abi_known_answers.c"),
                    os.path.join(built, "smatrix.so"), "-Wl,-rpath," + built, "-o", exe], check=True)


def test_router_binds_the_rccl_the_process_already_holds():
    """VERDICT r4 #7a: the C router binds RCCL at run time; a process that already has one mapped -- a torch process has torch's
    own librccl, on torch's HIP runtime -- must get THAT copy, not a second one from /opt/rocm (two RCCLs on two runtimes in one
    process).  Two fresh interpreters: without torch the loader's librccl.so.1 is taken; after `import torch` the path under
    torch/lib is.  No GPU needed: the library is only mapped and asked for its version."""
    import subprocess
    import sys
    code = "\n".join([
        "import ctypes as C, sys",
        "%s",
        "lib = C.CDLL(%r)",
        "lib.smatrix_shard_rccl_library.restype = C.c_char_p",
        "v = C.c_int(0)",
        "p = lib.smatrix_shard_rccl_library(C.byref(v))",
        "mapped = sorted({l.split()[-1] for l in open('/proc/self/maps') if 'librccl' in l})",
        "print(p.decode() if p else None, v.value, len(mapped))"])
    so = os.path.join(ROOT, "libsmatrix_amd", "lib", "smatrix.so")
    plain = subprocess.run([sys.executable, "-c", code % ("", so)], capture_output=True, text=True, timeout=300)
    assert plain.returncode == 0, plain.stderr[-2000:]
    path, version, n_mapped = plain.stdout.split()
    if path == "None":
        pytest.skip("no RCCL on this box")
    assert int(version) > 20000 and int(n_mapped) == 1
    with_torch = subprocess.run([sys.executable, "-c", code % ("import torch", so)], capture_output=True, text=True, timeout=600)
    assert with_torch.returncode == 0, with_torch.stderr[-2000:]
    tpath, tversion, tn_mapped = with_torch.stdout.split()
    torch_copy = any("librccl" in f for f in os.listdir(os.path.join(os.path.dirname(__import__("torch").__file__), "lib")))
    if torch_copy:
        assert "/torch/lib/" in tpath, (tpath, path)           # torch's copy, not a second one
    assert int(tn_mapped) == 1, "two RCCL copies mapped in one process"
    # SMATRIX_RCCL_LIB: a process that holds no RCCL yet loads the copy the variable names (here torch's, without importing torch)
    if torch_copy:
        tl = os.path.join(os.path.dirname(__import__("torch").__file__), "lib")
        cand = sorted(f for f in os.listdir(tl) if f.startswith("librccl.so"))
        named = subprocess.run([sys.executable, "-c", code % ("", so)], capture_output=True, text=True, timeout=300,
                               env=dict(os.environ, SMATRIX_RCCL_LIB=os.path.join(tl, cand[0])))
        assert named.returncode == 0, named.stderr[-2000:]
        npath, nversion, nn = named.stdout.split()
        assert "/torch/lib/" in npath and int(nn) == 1, (npath, path)
