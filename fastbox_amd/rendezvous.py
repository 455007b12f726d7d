"""Out-of-band hand-over of a few bytes from rank 0 to every other rank over TCP (no MPI, no torch.distributed): what
``fb_comm_create`` needs -- the 128-byte id of ``fb_comm_unique_id`` must reach every rank before the communicator exists.

Address and port come from the launcher's environment (MASTER_ADDR, MASTER_PORT as torch.distributed.run / the driver
set them; the port used here is MASTER_PORT + FASTBOX_RDV_OFFSET, default 17, so that it does not collide with the
launcher's own store)."""
import os
import socket
import struct
import time


def _endpoint(addr=None, port=None):
    addr = addr or os.environ.get("MASTER_ADDR", "127.0.0.1")
    if port is None:
        port = int(os.environ.get("MASTER_PORT", "29500")) + int(os.environ.get("FASTBOX_RDV_OFFSET", "17"))
    return addr, int(port)


def _recv_exact(conn, n):
    buf = b""
    while len(buf) < n:
        part = conn.recv(n - len(buf))
        if not part:
            raise ConnectionError("rendezvous: peer closed the connection")
        buf += part
    return buf


def broadcast_bytes(payload, rank, world, addr=None, port=None, timeout=120.0):
    """Rank 0 passes `payload` (bytes); every rank returns it.  Rank 0 serves world - 1 connections, the others connect
    (retrying until rank 0 listens or `timeout` seconds have passed)."""
    if world <= 1:
        return payload
    addr, port = _endpoint(addr, port)
    deadline = time.time() + timeout
    if rank == 0:
        srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
        srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
        srv.bind((addr if addr not in ("localhost",) else "127.0.0.1", port))
        srv.listen(world)
        srv.settimeout(timeout)
        seen = set()
        try:
            while len(seen) < world - 1:
                conn, _ = srv.accept()
                with conn:
                    conn.settimeout(timeout)
                    peer = struct.unpack("<i", _recv_exact(conn, 4))[0]
                    conn.sendall(struct.pack("<i", len(payload)) + payload)
                    seen.add(peer)
        finally:
            srv.close()
        return payload
    last = None
    while time.time() < deadline:
        try:
            with socket.create_connection((addr, port), timeout=5.0) as conn:
                conn.settimeout(timeout)
                conn.sendall(struct.pack("<i", rank))
                n = struct.unpack("<i", _recv_exact(conn, 4))[0]
                return _recv_exact(conn, n)
        except (ConnectionRefusedError, socket.timeout, OSError) as e:
            last = e
            time.sleep(0.2)
    raise TimeoutError("rendezvous with rank 0 at %s:%d failed: %r" % (addr, port, last))
