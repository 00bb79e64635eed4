"""Times the dense HunyuanVideo MLP leg of the C3 bench three ways: nn.Linear + GELU(tanh) + nn.Linear (the bench's form),
fc1 with the GELU folded into hipBLASLt's epilogue (torch._addmm_activation), and both under TunableOp."""
import os, sys, time
import torch

M, HID, FFN = int(os.environ.get("MLP_M", 119056)), 3072, 12288
dev = torch.device("cuda:0")
bf = dict(device=dev, dtype=torch.bfloat16)
g = torch.Generator(device=dev).manual_seed(1)
x = torch.randn(1, M, HID, generator=g, **bf)
layers = [(torch.nn.Linear(HID, FFN, **bf), torch.nn.Linear(FFN, HID, **bf)) for _ in range(6)]
act = torch.nn.GELU(approximate="tanh")


def plain(fc1, fc2):
    return fc2(act(fc1(x)))


def fused(fc1, fc2):
    h = torch._addmm_activation(fc1.bias, x.view(M, HID), fc1.weight.t(), use_gelu=True)
    return torch.addmm(fc2.bias, h, fc2.weight.t()).view(1, M, HID)


def timeit(fn, reps=3):
    with torch.no_grad():
        for fc1, fc2 in layers:
            fn(fc1, fc2)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            for fc1, fc2 in layers:
                fn(fc1, fc2)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / (reps * len(layers)) * 1e3


with torch.no_grad():
    a = plain(*layers[0]).float()
    b = fused(*layers[0]).float()
    print("max abs diff fused vs plain:", (a - b).abs().max().item(), "ref max", a.abs().max().item(), flush=True)
print("plain  ms/layer:", round(timeit(plain), 3), flush=True)
print("fused  ms/layer:", round(timeit(fused), 3), flush=True)
if os.environ.get("PROBE_TUNE", "1") == "1":
    import torch.cuda.tunable as tn
    tn.enable(True)
    tn.set_max_tuning_duration(int(os.environ.get("TUNE_MS", 30)))
    tn.set_max_tuning_iterations(10)
    tn.set_filename(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "tunableop_mlp.csv"))
    t0 = time.perf_counter()
    print("tuned plain ms/layer:", round(timeit(plain), 3), flush=True)
    print("tuned fused ms/layer:", round(timeit(fused), 3), "tuning took", round(time.perf_counter() - t0, 1), "s", flush=True)
    tn.write_file()
