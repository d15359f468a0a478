"""Process-group bootstrap without MPI and without PyTorch: a rank-0 TCP hub over
which the ranks of one node exchange the 128-byte RCCL unique id, meet at host-side barriers and reduce a
few scalars.  The data path never goes through here: filters are gathered by RCCL over xGMI
(`Engine.allgather_filters_dev`).

    rz = Rendezvous.from_env()            # RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT, as torchrun sets them
    uid = rz.broadcast(Engine.comm_unique_id() if rz.rank == 0 else None)
    eng.comm_init(uid, rz.rank, rz.world)

The hub listens on MASTER_PORT + APV_RDZV_PORT_OFFSET (default 17): the launcher's own store owns MASTER_PORT.
"""
import os
import pickle
import socket
import struct
import time


def _send(sock, obj):
    data = pickle.dumps(obj)
    sock.sendall(struct.pack("<Q", len(data)) + data)


def _recv(sock):
    hdr = b""
    while len(hdr) < 8:
        chunk = sock.recv(8 - len(hdr))
        if not chunk:
            raise ConnectionError("rendezvous peer closed the connection")
        hdr += chunk
    n = struct.unpack("<Q", hdr)[0]
    buf = bytearray()
    while len(buf) < n:
        chunk = sock.recv(min(1 << 16, n - len(buf)))
        if not chunk:
            raise ConnectionError("rendezvous peer closed the connection")
        buf += chunk
    return pickle.loads(bytes(buf))


class Rendezvous:
    def __init__(self, rank, world, addr="127.0.0.1", port=29617, timeout=120.0):
        self.rank, self.world = int(rank), int(world)
        self.peers = []          # rank 0: sockets of ranks 1..world-1, by rank
        self.hub = None          # other ranks: socket to rank 0
        if self.world == 1:
            return
        if self.rank == 0:
            srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
            srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            srv.bind((addr, port))
            srv.listen(self.world)
            srv.settimeout(timeout)
            by_rank = {}
            while len(by_rank) < self.world - 1:
                conn, _ = srv.accept()
                conn.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                conn.settimeout(timeout)
                by_rank[_recv(conn)] = conn
            srv.close()
            self.peers = [by_rank[r] for r in range(1, self.world)]
        else:
            deadline = time.time() + timeout
            while True:
                try:
                    s = socket.create_connection((addr, port), timeout=5.0)
                    break
                except OSError:
                    if time.time() > deadline:
                        raise
                    time.sleep(0.05)
            s.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
            s.settimeout(timeout)
            _send(s, self.rank)
            self.hub = s

    @classmethod
    def from_env(cls, **kw):
        rank = int(os.environ.get("RANK", "0"))
        world = int(os.environ.get("WORLD_SIZE", "1"))
        addr = os.environ.get("MASTER_ADDR", "127.0.0.1")
        port = int(os.environ.get("MASTER_PORT", "29600")) + int(os.environ.get("APV_RDZV_PORT_OFFSET", "17"))
        return cls(rank, world, addr, port, **kw)

    def gather(self, obj):
        """Rank 0 receives [obj of rank 0, ..., obj of rank world-1]; the others receive None."""
        if self.world == 1:
            return [obj]
        if self.rank == 0:
            return [obj] + [_recv(p) for p in self.peers]
        _send(self.hub, obj)
        return None

    def broadcast(self, obj):
        """Every rank receives rank 0's obj."""
        if self.world == 1:
            return obj
        if self.rank == 0:
            for p in self.peers:
                _send(p, obj)
            return obj
        return _recv(self.hub)

    def allreduce(self, value, op=max):
        vals = self.gather(value)
        return self.broadcast(op(vals) if self.rank == 0 else None)

    def barrier(self):
        self.allreduce(0)

    def close(self):
        for p in self.peers:
            p.close()
        if self.hub is not None:
            self.hub.close()
        self.peers, self.hub = [], None
