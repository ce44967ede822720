import os, subprocess, sys, tempfile, shutil
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bench_write_tree as bwt
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
top = tempfile.mkdtemp(prefix="tl_", dir="/dev/shm")
wt = os.path.join(top, "wt"); os.makedirs(wt)
print(bwt.make_c2(wt, 1000000), flush=True)
vit = os.path.join(ROOT, "vit_amd", "bin", "vit")
for mb in (128, 256, 512, 1024):
    env = dict(os.environ, VIT_DEVICES="0", VIT_DEVICE_DEFLATE="1", VIT_BATCH_MB=str(mb))
    shutil.rmtree(os.path.join(wt, ".git"), ignore_errors=True)
    subprocess.check_call([vit, "init"], cwd=wt, stdout=subprocess.DEVNULL, env=env)
    p = subprocess.run([vit, "write-tree", "--stats"], cwd=wt, capture_output=True, env=env)
    print(mb, p.stderr.decode().strip().splitlines()[-1], flush=True)
shutil.rmtree(top, ignore_errors=True)
