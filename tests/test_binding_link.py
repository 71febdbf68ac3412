"""CPU, build container only: the reference's UNCHANGED JNI glue (src/smatrix_jni.c) and Ruby glue
(src/smatrix_ruby.c) compile against test-double jni.h / ruby.h and link against this repo's smatrix.o
exactly as the reference's src/java/Makefile:22-23 and src/ruby/Makefile:18-19 do (glue + ../smatrix.o,
no -l flags) -- SURVEY.md 8f #4.  Skipped where /root/reference is absent (the GPU box)."""
import ctypes
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/src"
LIB = os.path.join(ROOT, "libsmatrix_amd", "lib")


@pytest.mark.skipif(not os.path.exists(os.path.join(REF, "smatrix_jni.c")), reason="reference sources absent")
def test_jni_glue_links_unchanged(tmp_path):
    if not os.path.exists(os.path.join(LIB, "smatrix.o")):
        subprocess.run(["make", "-C", os.path.join(ROOT, "libsmatrix_amd", "csrc")], check=True)
    so = str(tmp_path / "smatrix_java.so")
    subprocess.run(["gcc", "-shared", "-fPIC", "-w", "-I" + os.path.join(ROOT, "tests", "stubs"), "-I" + REF,
                    os.path.join(REF, "smatrix_jni.c"), os.path.join(LIB, "smatrix.o"), "-o", so], check=True)
    nm = subprocess.run(["nm", "-D", so], check=True, capture_output=True, text=True).stdout
    defined = {ln.split()[-1] for ln in nm.splitlines() if " T " in ln}
    undefined = {ln.split()[-1].split("@")[0] for ln in nm.splitlines() if " U " in ln}
    # the eight natives of SparseMatrix.java + the eight public smatrix_* from OUR object
    for n in ("init", "close", "get", "set", "incr", "decr", "getRowNative", "getRowLength"):
        assert "Java_com_paulasmuth_libsmatrix_SparseMatrix_" + n in defined
    for n in ("open", "close", "get", "set", "incr", "decr", "rowlen", "getrow"):
        assert "smatrix_" + n in defined
    assert not [u for u in undefined if u.startswith("smatrix_")], undefined    # nothing private needed
    assert undefined <= {"dlopen", "dlsym", "dlerror", "getenv", "fprintf", "printf", "abort", "malloc", "free",
                         "stderr", "__stack_chk_fail", "__cxa_finalize", "_ITM_deregisterTMCloneTable",
                         "_ITM_registerTMCloneTable", "__gmon_start__", "fwrite", "fputs", "puts",
                         "__printf_chk", "__fprintf_chk"}, undefined   # libc + libdl only
    ctypes.CDLL(so)        # loads: every dependency resolves without libamdhip64 (the shim dlopens it later)


@pytest.mark.skipif(not os.path.exists(os.path.join(REF, "smatrix_ruby.c")), reason="reference sources absent")
def test_ruby_glue_links_unchanged(tmp_path):
    """src/ruby/Makefile:18-19: $(CC) ... ../smatrix_ruby.c ../smatrix.o -o smatrix_ruby.so, the interpreter's
    rb_* symbols left undefined for the loader (-undefined dynamic_lookup there, the ELF default here)"""
    if not os.path.exists(os.path.join(LIB, "smatrix.o")):
        subprocess.run(["make", "-C", os.path.join(ROOT, "libsmatrix_amd", "csrc")], check=True)
    so = str(tmp_path / "smatrix_ruby.so")
    subprocess.run(["gcc", "-shared", "-fPIC", "-w", "-I" + os.path.join(ROOT, "tests", "stubs"), "-I" + REF,
                    os.path.join(REF, "smatrix_ruby.c"), os.path.join(LIB, "smatrix.o"), "-o", so], check=True)
    nm = subprocess.run(["nm", "-D", so], check=True, capture_output=True, text=True).stdout
    defined = {ln.split()[-1] for ln in nm.splitlines() if " T " in ln}
    undefined = {ln.split()[-1].split("@")[0] for ln in nm.splitlines() if " U " in ln}
    for n in ("Init_smatrix", "Init_smatrix_ruby", "smatrix_rb_initialize", "smatrix_rb_get", "smatrix_rb_set",
              "smatrix_rb_incr", "smatrix_rb_decr", "smatrix_rb_free", "smatrix_rb_gethandle"):
        assert n in defined
    # smatrix_ruby.c:33,37,73,100,127,154,163 call open/get/set/incr/decr/close: all come from OUR object
    for n in ("open", "close", "get", "set", "incr", "decr", "rowlen", "getrow"):
        assert "smatrix_" + n in defined
    assert not [u for u in undefined if u.startswith("smatrix_")], undefined
    ruby = {u for u in undefined if u.startswith("rb_")}
    assert ruby == {"rb_type", "rb_iv_get", "rb_iv_set", "rb_raise", "rb_define_class", "rb_define_method",
                    "rb_test_string_ptr", "rb_int2inum", "rb_num2int", "rb_data_object_wrap", "rb_data_object_get",
                    "rb_cObject", "rb_eTypeError"}, ruby
