import importlib, os, sys, time
sys.path.insert(0, os.getcwd())
import torch
par = importlib.import_module("pixel-art-raytracer_amd"); T = par.types
for name in ("512", "graybox"):
    if name == "512":
        W = H = 512; p = T.default_params(W, H, 512); a, l = par.scene_synthetic(64, W, H, 512, 12345)
    else:
        W, H = 480, 320; p = T.default_params(); a, l = par.scene_graybox(), T.make_light(480, 160, 80)
    r = par.Renderer(p, 0); r.set_scene(a, par.tile_floor(), l)
    fb = torch.zeros(W * H * 4, dtype=torch.uint8, device="cuda"); pal = torch.zeros(W * H, dtype=torch.uint8, device="cuda")
    ptrs = {"fb": fb.data_ptr(), "palidx": pal.data_ptr()}
    s = torch.cuda.current_stream().cuda_stream
    for _ in range(300): r.render_device(ptrs, stream=s)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(2000): r.render_device(ptrs, stream=s)
    torch.cuda.synchronize()
    print(name, "alone us/frame", (time.perf_counter() - t0) / 2000 * 1e6)
