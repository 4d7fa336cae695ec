"""ASan + UBSan over the CPU code (host mirror + oracle): the only sanitizer run this pool allows
(GPU AddressSanitizer / XNACK builds are refused), SURVEY.md §5."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.timeout(600)
def test_host_and_oracle_under_asan_ubsan(tmp_path):
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    exe = str(tmp_path / "san_driver")
    host = os.path.join(ROOT, "acgpathtracing_amd", "host")
    srcs = [os.path.join(ROOT, "tests", "san_driver.cpp"), os.path.join(ROOT, "oracle", "oracle_pt.cpp")] + \
           [os.path.join(host, f) for f in ("TinyObjWrapper.cpp", "Camera.cpp", "Trackball.cpp", "ImageIO.cpp")]
    cmd = ["g++", "-std=c++14", "-O1", "-g", "-ffp-contract=off", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
           "-fno-omit-frame-pointer", "-pthread", "-o", exe] + srcs
    subprocess.run(cmd, check=True, capture_output=True, text=True)
    scenes = os.path.join(ROOT, "acgpathtracing_amd", "scenes")
    golden = os.path.join(ROOT, "tests", "golden", "obj")
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([exe, os.path.join(scenes, "cornell_box.obj"), str(tmp_path), os.path.join(golden, "quads_ngons.obj"),
                        os.path.join(golden, "no_mtl.obj"), os.path.join(golden, "exponent_numbers.obj")],
                       capture_output=True, text=True, env=env, timeout=500)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "SANITIZED_RUN_OK" in r.stdout
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr
