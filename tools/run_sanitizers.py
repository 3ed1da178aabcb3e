#!/usr/bin/env python3
"""The product's own kernels + host scheduler (tests/hostsim: the unchanged sources of libpob_hip.so on the HIP-on-fibers shim) and the C oracle under
AddressSanitizer + UndefinedBehaviorSanitizer: builds libpob_hostsim_san.so / oracle/liboracle_san.so with clang and runs the CPU-shim tests in a
process that has clang's sanitizer runtime preloaded.  The kernels' deliberate past-the-slab buffer offsets are MODELLED by the shim (a raw-buffer load
there returns 0, a store is dropped -- what the hardware does), not hidden: anything else out of bounds is an error.
    python tools/run_sanitizers.py [quick|full] [> profiles/roundN_sanitizers.txt]"""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLANG = "/opt/rocm/lib/llvm/bin/clang"
QUICK = "spend_suite or run_shim or inverse_paths"
# (the loaded library is checked too: a run that silently used the uninstrumented build would prove nothing)
FULL = ("spend_suite or fixture_suite or pokes_in_every_class_spend or service_loop or reference_suites_on_the_shim or run_shim or gadget_mains_evaluator or "
        "gadget_mains_seeded or production_sizes or inverse_paths or pipelined or selfcheck or inorder or riding or records_of_the_evaluation")


def main(mode="quick"):
    rt = subprocess.check_output([CLANG, "-print-file-name=libclang_rt.asan-x86_64.so"], text=True).strip()
    env = dict(os.environ, LD_PRELOAD=rt, POB_HOSTSIM_SAN="1", ORACLE_SAN="1",
               ASAN_OPTIONS="detect_leaks=0:halt_on_error=1:abort_on_error=0:detect_stack_use_after_return=0",
               UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    t0 = time.time()
    r = subprocess.run([sys.executable, "-m", "pytest", "tests/test_hostsim_cpu.py", "-x", "-q", "-p", "no:cacheprovider", "-k", QUICK if mode == "quick" else FULL],
                       cwd=ROOT, env=env, capture_output=True, text=True)
    out = r.stdout + r.stderr
    for lib in ("tests/hostsim/libpob_hostsim_san.so", "oracle/liboracle_san.so"):
        if not os.path.exists(os.path.join(ROOT, lib)):
            print(f"{lib} was not built: the run did not use the sanitizer builds"); return 1
    findings = [ln for ln in out.splitlines() if "ERROR: AddressSanitizer" in ln or "runtime error:" in ln or "SUMMARY: " in ln]
    print(f"sanitizers ({mode}): -fsanitize=address,undefined on tests/hostsim/libpob_hostsim_san.so + oracle/liboracle_san.so, runtime {os.path.basename(rt)}")
    print(f"pytest rc {r.returncode} in {time.time() - t0:.0f} s; findings: {len(findings)}")
    for ln in findings[:40]:
        print("  " + ln)
    print(out[-3000:])
    return 0 if r.returncode == 0 and not findings else 1


if __name__ == "__main__":
    sys.exit(main(*(sys.argv[1:2] or ["quick"])))
