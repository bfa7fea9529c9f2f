"""Worker for test_rccl_rendezvous_under_torchrun: runs `Communicator.from_environment()` exactly as
bench.py does under torch.distributed.run, with the two RCCL calls replaced by recorders (no GPU
here), and reports which unique id each rank ended up with."""

import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from bodge_amd import solver  # noqa: E402


def main():
    out_dir = sys.argv[1]
    seen = {}

    def fake_unique_id():
        return bytes([int(os.environ["RANK"]) + 1]) * 64 + os.urandom(64)

    def fake_init(self, rank, n_ranks, device, unique_id):
        # like ncclCommInitRank, returns only once every rank has arrived with the id
        import time

        self.rank, self.n_ranks, self.device = rank, n_ranks, device
        self._handle = None
        seen["uid"] = unique_id
        open(os.path.join(out_dir, f"arrived{rank}"), "w").close()
        deadline = time.time() + 120
        while not all(os.path.exists(os.path.join(out_dir, f"arrived{r}")) for r in range(n_ranks)):
            if time.time() > deadline:
                raise RuntimeError("fake communicator init timed out")
            time.sleep(0.02)

    solver.Communicator.new_unique_id = staticmethod(fake_unique_id)
    solver.Communicator.__init__ = fake_init
    solver.Communicator.barrier = lambda self: None
    solver.Communicator.close = lambda self: None
    comm = solver.Communicator.from_environment(timeout=60)
    with open(os.path.join(out_dir, f"rank{comm.rank}.json"), "w") as fh:
        json.dump({"rank": comm.rank, "world": comm.n_ranks, "device": comm.device, "ppid": os.getppid(),
                   "uid": hashlib.sha256(seen["uid"]).hexdigest(), "len": len(seen["uid"])}, fh)


if __name__ == "__main__":
    main()
