"""Fixed cost of a timed block: time of K frames (4 in flight, started from an idle device and waited for) for
several K and several ways of waiting; a + b*K fit per way. Run on the GPU box."""
import importlib, os, statistics, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
par = importlib.import_module("pixel-art-raytracer_amd")
pipeline = importlib.import_module("pixel-art-raytracer_amd.pipeline")
T = importlib.import_module("pixel-art-raytracer_amd.types")

def main():
    depth = int(os.environ.get("DEPTH", "4"))
    W = 4096
    params = T.default_params(W, W, W)
    aabbs, light = par.scene_synthetic(1024, W, W, W, 12345)
    pipe = pipeline.FramePipeline(params, aabbs, par.tile_floor(), light, depth=depth)
    def wait_sync():
        torch.cuda.synchronize()
    def wait_spin():
        for s in pipe.slots:
            while not s.stream.query():
                pass
        torch.cuda.synchronize()
    def wait_streams():
        for s in pipe.slots:
            s.stream.synchronize()
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 1.0:
        pipe.submit_many(0, 64); torch.cuda.synchronize()
    for name, wait in (("device synchronize", wait_sync),):
        xs, ys = [], []
        for K in (4, 8, 20, 50, 200, 1000):
            ts = []
            for _ in range(30):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                pipe.submit_many(0, K)
                wait()
                ts.append((time.perf_counter() - t0) * 1e6)
            med = statistics.median(ts)
            xs.append(K); ys.append(med)
            print(f"{name:20s} K={K:5d}  block {med:9.1f} us  per frame {med / K:7.2f}  min {min(ts) / K:7.2f}", flush=True)
        b, a = np.polyfit(xs, ys, 1)
        print(f"{name:20s} fit: {a:.1f} us + {b:.2f} us x K", flush=True)
    pipe.close()
main()
