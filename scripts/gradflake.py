import sys, os, io, contextlib
sys.path[:0] = ["/root/repo", "/root/repo/tests"]
os.chdir(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sys.path[:0] = [os.getcwd(), os.path.join(os.getcwd(), "tests")]
import test_gpu_train as T
fails = 0
for i in range(30):
    buf = io.StringIO()
    try:
        with contextlib.redirect_stdout(buf):
            T.test_gradients_vs_reference_f64()
    except AssertionError as e:
        fails += 1
        print("FAIL", i, str(e)[:400].replace("\n", " | "))
    print(i, buf.getvalue().strip()[:230])
print("fails", fails)
