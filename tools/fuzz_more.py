"""Extended run of the randomised GPU parity tests (more seeds than the test-suite uses)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import test_gpu_fuzz as F
a, b = int(sys.argv[1]), int(sys.argv[2])
t0 = time.time()
for seed in range(a, b):
    F.test_fuzz_batch.__wrapped__(seed) if hasattr(F.test_fuzz_batch, "__wrapped__") else F.test_fuzz_batch(seed)
    if seed % 3 == 0:
        F.test_fuzz_streaming(seed)
    if seed % 20 == 0:
        print(f"seed {seed} ok ({time.time() - t0:.0f} s)", flush=True)
print("all ok", a, b)
