"""Key-value rendezvous between the ranks of one launch, over TCP (standard library only).

What it is for: before `ncclCommInitRank` every rank needs rank 0's 128-byte RCCL unique id,
and a launcher script may want a few scalars agreed between ranks before any GPU call.  Rank 0
serves a small dictionary on `MASTER_ADDR`, the other ranks are clients; `get` blocks until the
key exists.  Nothing touches the file system, so there is nothing stale to find after a crashed
run, and the ranks may sit on different nodes.  The server binds to MASTER_ADDR only (127.0.0.1 for
a one-node launch), keys are write-once (a second `set` of an existing key is refused), and every
message carries the launch's nonce - which keeps launches apart, but is NOT a secret (run id, port
and parent pid can be read by any local user): this is a rendezvous among cooperating processes of
one user on a trusted node, not an authenticated channel.

Port: the launcher's own store owns MASTER_PORT, so this one listens on a port derived from it
(MASTER_PORT + 1 + k for the first k in 0..31 that can be bound).  Every request carries the
launch's nonce (TORCHELASTIC_RUN_ID, MASTER_PORT and - on one node - the launcher's pid); a
server that belongs to another launch answers "wrong launch" and the client moves on to the next
candidate port.  Messages are single JSON lines, values hex-encoded bytes.
"""

from __future__ import annotations

import json
import os
import socket
import socketserver
import threading
import time

PORT_CANDIDATES = 32


def launch_nonce() -> str:
    run_id = os.environ.get("TORCHELASTIC_RUN_ID", "none")
    parent = os.getppid() if int(os.environ.get("GROUP_WORLD_SIZE", os.environ.get("NNODES", "1")) or 1) <= 1 else 0
    return f"{run_id}/{os.environ.get('MASTER_PORT', '0')}/{parent}"


class _Handler(socketserver.StreamRequestHandler):
    def handle(self):
        server = self.server
        for line in self.rfile:
            try:
                msg = json.loads(line)
            except ValueError:
                return
            if msg.get("nonce") != server.nonce:
                reply = {"ok": False, "error": "wrong launch"}
            elif msg.get("op") == "set":
                with server.changed:
                    fresh = msg["key"] not in server.table  # write-once: nobody overwrites an id or a gathered value
                    if fresh:
                        server.table[msg["key"]] = msg["value"]
                        server.changed.notify_all()
                reply = {"ok": fresh, "error": "" if fresh else "key already set"}
            elif msg.get("op") == "get":
                deadline = time.time() + float(msg.get("timeout", 300.0))
                with server.changed:
                    while msg["key"] not in server.table and time.time() < deadline:
                        server.changed.wait(min(1.0, max(0.0, deadline - time.time())))
                    value = server.table.get(msg["key"])
                reply = {"ok": value is not None, "value": value, "error": "timed out" if value is None else ""}
            else:
                reply = {"ok": msg.get("op") == "hello"}
            self.wfile.write((json.dumps(reply) + "\n").encode())
            self.wfile.flush()


class _Server(socketserver.ThreadingTCPServer):
    allow_reuse_address = True
    daemon_threads = True


class Store:
    """`Store(rank, world)`: rank 0 starts the server thread; everyone gets a client."""

    def __init__(self, rank: int, world: int, addr: str | None = None, base_port: int | None = None,
                 nonce: str | None = None, timeout: float = 300.0):
        self.rank, self.world = rank, world
        self.addr = addr or os.environ.get("MASTER_ADDR", "127.0.0.1")
        self.base_port = int(base_port if base_port is not None else os.environ.get("MASTER_PORT", "29500")) + 1
        self.nonce = nonce or launch_nonce()
        self.timeout = timeout
        self._server = None
        self._sock = None
        self._io = None
        self._lock = threading.Lock()
        self._counter = 0
        if rank == 0:
            self._serve()
        self._connect()

    def _serve(self) -> None:
        last = None
        for k in range(PORT_CANDIDATES):
            try:
                server = _Server((self._bind_address(), self.base_port + k), _Handler)
            except OSError as exc:
                last = exc
                continue
            server.nonce, server.table, server.changed = self.nonce, {}, threading.Condition()
            threading.Thread(target=server.serve_forever, daemon=True, name="bodge-amd-rendezvous").start()
            self._server = server
            return
        raise RuntimeError(f"rendezvous: no free port in {self.base_port}..{self.base_port + PORT_CANDIDATES - 1}: {last}")

    def _bind_address(self) -> str:
        """The interface MASTER_ADDR names; all interfaces only if it does not resolve to a local one."""
        try:
            probe = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
            try:
                probe.bind((self.addr, 0))
            finally:
                probe.close()
            return self.addr
        except OSError:
            return ""

    def _connect(self) -> None:
        deadline = time.time() + self.timeout
        while True:
            for k in range(PORT_CANDIDATES):
                try:
                    sock = socket.create_connection((self.addr, self.base_port + k), timeout=0.5 if self.addr.startswith("127.") else 2.0)
                except OSError:
                    continue
                sock.settimeout(None)
                io = sock.makefile("rwb")
                try:
                    io.write((json.dumps({"nonce": self.nonce, "op": "hello"}) + "\n").encode())
                    io.flush()
                    reply = json.loads(io.readline() or b"{}")
                except (OSError, ValueError):
                    reply = {}
                if reply.get("ok"):
                    self._sock, self._io = sock, io
                    return
                sock.close()
            if time.time() > deadline:
                raise RuntimeError(f"rank {self.rank}: no rendezvous server of this launch answered at "
                                   f"{self.addr}:{self.base_port}+ within {self.timeout:.0f} s")
            time.sleep(0.05)

    def _call(self, msg: dict) -> dict:
        msg["nonce"] = self.nonce
        with self._lock:
            self._io.write((json.dumps(msg) + "\n").encode())
            self._io.flush()
            line = self._io.readline()
        if not line:
            raise RuntimeError("rendezvous: the server closed the connection")
        return json.loads(line)

    def set(self, key: str, value) -> None:
        data = value.encode() if isinstance(value, str) else bytes(value)
        if not self._call({"op": "set", "key": key, "value": data.hex()}).get("ok"):
            raise RuntimeError(f"rendezvous: set({key}) refused")

    def get(self, key: str, timeout: float | None = None) -> bytes:
        reply = self._call({"op": "get", "key": key, "timeout": self.timeout if timeout is None else timeout})
        if not reply.get("ok"):
            raise RuntimeError(f"rank {self.rank}: rendezvous get({key}): {reply.get('error')}")
        return bytes.fromhex(reply["value"])

    # ---- collectives over the store: a handful of scalars between a handful of ranks
    def gather(self, payload: bytes) -> list[bytes]:
        """Every rank's payload, in rank order (doubles as a barrier)."""
        self._counter += 1
        tag = f"gather{self._counter}"
        self.set(f"{tag}/{self.rank}", payload)
        return [self.get(f"{tag}/{r}") for r in range(self.world)]

    def barrier(self) -> None:
        self.gather(b"")

    def finish(self, timeout: float = 60.0) -> None:
        """Last call of a launch: rank 0 keeps serving until every rank has signed off."""
        self.set(f"bye/{self.rank}", b"")
        if self.rank == 0:
            for r in range(self.world):
                self.get(f"bye/{r}", timeout)


_store: Store | None = None


def store_from_environment(timeout: float = 300.0) -> Store:
    """The store of this `torch.distributed.run`-style launch (RANK / WORLD_SIZE / MASTER_*), created once."""
    global _store
    if _store is None:
        _store = Store(int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), timeout=timeout)
    return _store
