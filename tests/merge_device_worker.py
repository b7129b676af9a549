"""One rank of the 2-rank test of merge_device_maps on a one-GPU box (tests/test_gpu_boundary.py): both ranks' maps live on
GPU 0, packing / finishing / importing run as the library's HIP kernels, the collectives go over gloo (host staged).
usage: merge_device_worker.py RANK WORLD PORT OUT_DIR"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def note(rank, what):
    print(f"[merge worker {rank}] {what}", file=sys.stderr, flush=True)


def dump(m, path):
    b = m.export_blocks()
    np.savez(path, keys=b["keys"], log_odds=b["log_odds"], occ=b["occ"])


def main():
    rank, world, port, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
    import torch
    import torch.distributed as dist

    from mlmapping_amd import synthetic as syn
    from mlmapping_amd.config import SDEF
    from mlmapping_amd.merge import merge_device_maps
    from mlmapping_amd.mlmap import MLMap

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = port
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg = SDEF
    m = MLMap(cfg, device=0, max_blocks=4096)
    frames = list(syn.stream(cfg, "room_jitter", "random", 6, seed=42 + rank))
    for img, (q, t) in frames[:3]:
        m.update_map(img, q, t)
        note(rank, f"frame integrated: {m.frame_stats()}")
    dump(m, os.path.join(out, f"own1_{rank}.npz"))
    note(rank, "first merge")
    merge_device_maps(m, load_back=True)
    dump(m, os.path.join(out, f"merged1_{rank}.npz"))
    note(rank, "merged map loaded back")
    for img, (q, t) in frames[3:]:
        m.update_map(img, q, t)
        note(rank, f"frame integrated: {m.frame_stats()}")
    dump(m, os.path.join(out, f"own2_{rank}.npz"))
    note(rank, "second merge")
    merge_device_maps(m, load_back=True)
    dump(m, os.path.join(out, f"merged2_{rank}.npz"))
    merge_device_maps(m, load_back=True)  # nothing new observed: unchanged
    dump(m, os.path.join(out, f"merged3_{rank}.npz"))
    note(rank, "done")
    dist.barrier()
    dist.destroy_process_group()
    m.close()


if __name__ == "__main__":
    main()
