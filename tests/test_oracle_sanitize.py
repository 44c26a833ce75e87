"""The CPU oracle under AddressSanitizer + UBSan (CPU build only; GPU sanitizers are not
available on this pool)."""
import os
import subprocess

from conftest import GOLDEN, ROOT


def test_oracle_under_asan_ubsan(tmp_path):
    exe = str(tmp_path / "ox_san")
    subprocess.check_call(["gcc", "-O1", "-g", "-std=c11", "-pthread", "-fsanitize=address,undefined",
                           "-fno-omit-frame-pointer", "-fno-sanitize-recover=undefined", "-o", exe,
                           os.path.join(ROOT, "oracle", "redux_oracle.c"), os.path.join(ROOT, "oracle", "sanitize_main.c")])
    out = subprocess.run([exe, os.path.join(GOLDEN, "corpora", "canterbury", "alice29.txt")], capture_output=True,
                         text=True, timeout=300)
    assert out.returncode == 0 and "sanitize ok" in out.stdout, out.stdout + out.stderr
