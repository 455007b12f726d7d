"""CPU: the out-of-band hand-over of the RCCL unique id (fastbox_amd.rendezvous): rank 0 serves 128 bytes to the other
ranks over TCP; latecomers and early birds both get them."""
import multiprocessing as mp
import socket
import time


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _rank(rank, world, port, delay, q):
    from fastbox_amd.rendezvous import broadcast_bytes
    time.sleep(delay)
    payload = bytes(range(128)) if rank == 0 else None
    q.put((rank, broadcast_bytes(payload, rank, world, "127.0.0.1", port, timeout=30.0)))


def test_unique_id_reaches_every_rank():
    world, port = 4, _free_port()
    q = mp.Queue()
    # rank 0 starts late (the others retry), rank 3 later still (rank 0 keeps serving)
    ps = [mp.Process(target=_rank, args=(r, world, port, {0: 0.5, 3: 1.0}.get(r, 0.0), q)) for r in range(world)]
    for p in ps:
        p.start()
    got = dict(q.get(timeout=60) for _ in range(world))
    for p in ps:
        p.join(timeout=30)
        assert p.exitcode == 0
    assert all(got[r] == bytes(range(128)) for r in range(world))


def test_one_rank_needs_no_socket():
    from fastbox_amd.rendezvous import broadcast_bytes
    assert broadcast_bytes(b"abc", 0, 1) == b"abc"
