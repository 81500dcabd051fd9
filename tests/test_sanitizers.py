"""CPU sanitizer leg: the product's host-compiled code -- csrc/host_entry.h
(subset stream, replay of RANSAC.hxx:49-117, duplicate set, single-datum agree / estimate), csrc/lm_core.h and the
per-model headers, the plugin loop of lsqrrecipes_amd/include/RANSAC.h -- and the oracle's C sources built with
-fsanitize=address,undefined -fno-sanitize-recover and driven by tests/sanitize/driver.cpp (exact-size heap buffers,
non-power-of-two dense dimensions, duplicate-ridden batches).  Sanitizers run on the CPU build only; no GPU involved."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SAN = ["-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer", "-g", "-O1",
       "-ffp-contract=off"]


def test_host_code_under_address_and_undefined_behaviour_sanitizers(tmp_path):
    from lsqrrecipes_amd import _lib
    _lib.load()   # the shared library the device-side symbols of RANSAC.h resolve against must exist
    objs = []
    for src in ("linalg.c", "estimators.c", "ransac.c"):
        o = str(tmp_path / (src + ".o"))
        subprocess.check_call(["gcc", "-std=c11", "-c", *SAN, "-I", os.path.join(ROOT, "oracle"),
                               "-o", o, os.path.join(ROOT, "oracle", src)])
        objs.append(o)
    exe = str(tmp_path / "san_driver")
    subprocess.check_call(
        ["g++", "-std=c++20", *SAN, "-Wall", "-Wno-attributes", "-Wno-unused-function",
         "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "lsqrrecipes_amd", "include"),
         "-I", os.path.join(ROOT, "lsqrrecipes_amd", "csrc"), "-I", os.path.join(ROOT, "oracle"),
         "-o", exe, os.path.join(ROOT, "tests", "sanitize", "driver.cpp"), *objs,
         "-L", os.path.join(ROOT, "lsqrrecipes_amd"), "-llsqr_hip",
         "-Wl,-rpath," + os.path.join(ROOT, "lsqrrecipes_amd"), "-lm"])
    supp = tmp_path / "lsan.supp"
    supp.write_text("leak:libamdhip64\nleak:libhsa-runtime64\nleak:librocprofiler\nleak:libamd_comgr\n")
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0:strict_string_checks=1",
               UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1",
               LSAN_OPTIONS="suppressions=%s:print_suppressions=0" % supp)
    r = subprocess.run([exe], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert r.returncode == 0 and "sanitizer driver ok" in r.stdout, r.stdout[-4000:]
