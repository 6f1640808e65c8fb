"""tests/virtual_ranks.py — N ranks of a decomposed run as N stepper handles in ONE process on one GPU.

Every rank is a real `csim_stepper` of the product in `external_halo` mode on its own tile of the
`csim_decomp_init` decomposition; the faces (depth 1: the four edge lines, reference src/halo.cpp:28-46;
depth 2..7: the deep faces of a fused pass in 8 directions incl. the diagonal corner blocks) are routed between
the handles by this process — no process cap, no RCCL, so the 4 x 2 topology of the 8-GPU runs (reference
src/decomp.cpp:13-33: mid-x ranks with three side and two diagonal peers) can be assembled at full size on a
one-GPU box.  What is NOT covered here is the transport itself (RCCL over xGMI)."""
import numpy as np

OPPOSITE4 = {0: 1, 1: 0, 2: 3, 3: 2}


def opposite8(d):
    return d ^ 1 if d < 4 else 11 - d


class VirtualRanks:
    def __init__(self, csim, world, nx, ny, dx=1.0, dy=1.0, bc=(0, 0, 0, 0), bc_value=0.0, fuse=None):
        self.csim, self.world = csim, world
        self.decs = [csim.decomp_init(world, r, nx, ny) for r in range(world)]
        self.st = []
        for dec in self.decs:
            st = csim.Stepper(dec, dx, dy, bc, bc_value)
            st.set_option("external_halo", 1)
            if fuse is not None:
                st.set_option("fuse", fuse)
            self.st.append(st)
        limits = {st.fuse_limit() for st in self.st}
        assert len(limits) == 1, f"ranks disagree on the deepest pass: {limits}"
        self.depth = limits.pop()

    def close(self):
        for st in self.st:
            st.close()
        self.st = []

    def upload_tiles(self, tile_of_rank):
        """tile_of_rank(r, dec) -> (ny_local + 2, nx_local + 2) array with the interior filled"""
        for r, (dec, st) in enumerate(zip(self.decs, self.st)):
            st.upload(tile_of_rank(r, dec))

    def upload_global(self, interior):
        def tile(r, dec):
            u = np.zeros((dec.ny_local + 2, dec.nx_local + 2))
            u[1:-1, 1:-1] = interior[dec.y_offset:dec.y_offset + dec.ny_local, dec.x_offset:dec.x_offset + dec.nx_local]
            return u
        self.upload_tiles(tile)

    def _exchange_lines(self):
        lines = [st.halo_pack() for st in self.st]
        for r, (dec, st) in enumerate(zip(self.decs, self.st)):
            st.halo_unpack([lines[dec.nbr[k]][OPPOSITE4[k]] if dec.nbr[k] >= 0 else None for k in range(4)])

    def _exchange_faces(self, t):
        peers = [st.faces_neighbors(t)[0] for st in self.st]
        faces = [st.faces_pack(t) for st in self.st]
        for r, st in enumerate(self.st):
            # the face rank q packed for direction d goes to peers_q[d], which takes it as coming from opposite8(d)
            got = []
            for d in range(8):
                q = peers[r][d]
                if q < 0:
                    got.append(None)
                    continue
                assert peers[q][opposite8(d)] == r, (r, d, q)
                got.append(faces[q][opposite8(d)])
            st.faces_unpack(t, got)

    def advance(self, D, dt, vx, vy, nsteps, depth=None):
        """the schedule of host_transport.advance: fused passes while at least three steps remain, then single steps"""
        depth = self.depth if depth is None else min(depth, self.depth)
        remaining = nsteps
        while remaining >= 3 and depth >= 2:
            t = min(depth, remaining - 1)
            self._exchange_faces(t)
            for st in self.st:
                st.run(D, dt, vx, vy, t)
            remaining -= t
        while remaining > 0:
            self._exchange_lines()
            for st in self.st:
                st.run(D, dt, vx, vy, 1)
            remaining -= 1

    def download(self, r):
        return self.st[r].download()

    def checksum(self):
        return sum(st.checksum() for st in self.st) % (1 << 64)


def tile_mask(dec, corners=False):
    """cells of a rank's local array that the reference defines: interior + the ghost lines (span 1..n) of every
    side; the four corner ghosts only on request (undefined across ranks)"""
    m = np.ones((dec.ny_local + 2, dec.nx_local + 2), bool)
    if not corners:
        m[[0, 0, -1, -1], [0, -1, 0, -1]] = False
    return m


def physical_mask(dec):
    """interior + the ghost lines of PHYSICAL sides: what a slice of a single-field run defines for this tile
    (its neighbour-side ghost lines hold halo values of the state before the last step instead)"""
    m = np.zeros((dec.ny_local + 2, dec.nx_local + 2), bool)
    m[1:-1, 1:-1] = True
    phys = [n < 0 for n in dec.nbr]
    m[1:-1, 0] |= phys[0]
    m[1:-1, -1] |= phys[1]
    m[0, 1:-1] |= phys[2]
    m[-1, 1:-1] |= phys[3]
    return m
