"""Single-rank probe on real hardware: RCCL accepts the in-place all_gather_into_tensor layout used by
protstruc_amd.distributed._allgather_rows (input view == its own slot of the output)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
dist.init_process_group("nccl", device_id=dev)
from protstruc_amd.distributed import _allgather_rows, pairwise_distance_matrix_sharded
full = torch.randn(3, 8, 8, 15, 15, device=dev); ref = full.clone()
_allgather_rows(full, 0, 8, 0, 1, None)
_allgather_rows(full.view(torch.uint8).view(3, 8, -1)[:, :, :64].contiguous().view(3, 8, 64), 0, 8, 0, 1, None)
torch.cuda.synchronize()
print("in-place all_gather_into_tensor ok:", torch.equal(full, ref))
xyz = torch.randn(2, 32, 15, 3, device=dev); mask = torch.rand(2, 32, 15, device=dev) < 0.9
d, m, rng = pairwise_distance_matrix_sharded(xyz, mask, gather=True)
from protstruc_amd import ops
d0, m0 = ops.pairwise_distance(xyz, mask)
print("sharded == single:", torch.equal(d, d0) and torch.equal(m, m0), rng)
dist.destroy_process_group()
