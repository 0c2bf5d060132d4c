"""Worker of tests/test_dist_gloo.py: the N>1 path on CPU ranks (gloo) against the emulator library --
rank 0 loads the weights, ONE broadcast of the packed arena, every rank generates its own contiguous shard
of the request list with no further collective, results are gathered host-side."""
import os
import sys

import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, os.path.join(ROOT, "neutts-air_amd"), HERE):
    sys.path.insert(0, p)

from neutts import _hip, dist as ndist  # noqa: E402
from oracle import backbone_ref as br  # noqa: E402
from common import engine_cfg  # noqa: E402


def main():
    lib = sys.argv[1]
    backend = sys.argv[2] if len(sys.argv) > 2 else "gloo"     # "nccl" = RCCL: tests/test_gpu_dist.py on a box with >= 2 GPUs
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if backend == "nccl":
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    else:
        dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    cfg = br.BackboneConfig.tiny(vocab_size=512, num_layers=1)
    eng = _hip.BackboneEngine(engine_cfg(cfg, max_batch=2, max_context=64, max_prefill_tokens=128), local if backend == "nccl" else 0, lib)
    w = br.make_weights(cfg, 17, walk_gain=4.0)
    if rank == 0:
        eng.load_state_dict({k: v.numpy() for k, v in w.items()}, inv_freq=br.rope_inv_freq(cfg).numpy())
    ndist.broadcast_weights(eng, src=0, device=torch.device("cuda", local) if backend == "nccl" else torch.device("cpu"))
    prompts = [br.synthetic_prompt(cfg, 50 + i, 6 + 3 * i) for i in range(5)]
    lo, hi = ndist.shard_range(len(prompts), rank, world)
    eos = cfg.vocab_size - 1
    samp = [_hip.Sampling(max_length=len(p) + 6, min_new_tokens=6, eos_token_id=eos, do_sample=False) for p in prompts]
    mine = eng.generate(prompts[lo:hi], samp[lo:hi])
    allr = ndist.gather_results(mine, world)
    if rank == 0:
        wd = br.cast_weights(w, torch.bfloat16)
        want = [br.generate(cfg, wd, p, len(p) + 6, eos, min_new_tokens=6).ids for p in prompts]
        assert allr == want, (allr, want)
        print("DIST_OK", world, [len(x) for x in allr], flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
