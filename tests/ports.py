"""Rendezvous port for the multi-process tests.

bind(0) hands out a port from the kernel's EPHEMERAL range (ip_local_port_range, 32768-60999 here) — the same range
every outgoing connection of every other process draws its source port from.  Between closing that probe socket and
rank 0's TCPStore listening on the number, a gloo / RCCL bootstrap connection of an earlier test can take it: rank 0
then dies with EADDRINUSE and the other ranks wait for a store that never comes (seen once in ~300 GPU-box test runs:
profiles/r03_multirank_rendezvous_race.txt).  Ports BELOW the ephemeral range are only ever taken by an explicit
bind, so a number from there that binds now still binds a moment later."""
import os
import random
import socket


def rendezvous_port():
    lo = 20000
    try:
        with open("/proc/sys/net/ipv4/ip_local_port_range") as f:
            hi = min(int(f.read().split()[0]), 32768)
    except OSError:
        hi = 32768
    if hi - lo < 1000:   # an unusual range: fall back to its lower neighbourhood
        lo, hi = 10000, 20000
    rng = random.Random(os.getpid() * 1000003 + int.from_bytes(os.urandom(4), "little"))
    for _ in range(200):
        p = rng.randrange(lo, hi)
        s = socket.socket()
        try:
            s.bind(("127.0.0.1", p))
        except OSError:
            continue
        finally:
            s.close()
        return p
    raise RuntimeError("no free rendezvous port found")
