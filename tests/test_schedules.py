"""CPU model of the hand-over protocol of the DMA-ring kernels (no GPU needed).

The ping-pong kernels (gemm_hls_amd/csrc/mm_mfma_f16.hip, mm_mfma_i8.hip, the fp32 variant) and the
DMA-staged VALU kernel (mm_valu_tile.inc) order LDS traffic with nothing but counted `s_waitcnt
vmcnt(N)` and `s_barrier`.  A wrong count does not fail a test reliably -- the DMA usually lands in
time anyway -- so the protocol is checked here by construction: each wave's program is replayed as
a list of (epoch, action), epoch = number of workgroup barriers passed, and two rules are asserted
for every slab (cdna_hip_programming.md, 8-phase template):

  RAW  a slab is read only in an epoch strictly AFTER the epoch in which EVERY wave executed the
       wait that retires its own DMA pieces of that slab (the barrier between publishes them);
  WAR  a buffer is refilled only in an epoch strictly after the last epoch in which any wave read
       its previous occupant (those reads are complete: lgkmcnt(0) / consumed before the barrier).

vmcnt semantics used: DMA loads retire in issue order; `vmcnt(N)` returns when at most N are
outstanding, i.e. everything but the N most recently issued pieces has landed."""
import pytest


class Wave:
    def __init__(self):
        self.epoch = 0
        self.issued = []     # (what, pieces) in issue order
        self.reads = []      # (epoch, what)
        self.issues = []     # (epoch, what)
        self.retired = {}    # what -> epoch of the wait that covers it

    def barrier(self):
        self.epoch += 1

    def issue(self, what, pieces):
        self.issued.append((what, pieces))
        self.issues.append((self.epoch, what))

    def wait_vmcnt(self, n):
        outstanding = 0
        keep = 0
        for what, pieces in reversed(self.issued):   # the newest pieces may stay in flight
            if outstanding + pieces <= n:
                outstanding += pieces
                keep += 1
            else:
                break
        for what, _ in self.issued[:len(self.issued) - keep]:
            self.retired.setdefault(what, self.epoch)

    def read(self, what):
        self.reads.append((self.epoch, what))


def check(waves, buffer_of, what_label="slab"):
    every = set()
    for w in waves:
        every |= {what for _, what in w.reads}
    for what in every:
        published = 1 + max(w.retired.get(what, 10 ** 9) for w in waves)
        first_read = min(e for w in waves for e, x in w.reads if x == what)
        assert first_read >= published, f"RAW: {what_label} {what} read in epoch {first_read}, published in {published}"
    # WAR: an issue of X into buffer b must come after every read of the previous occupant of b
    for w in waves:
        for e_issue, what in w.issues:
            b = buffer_of(what)
            prev = [x for x in every if buffer_of(x) == b and x < what]
            if not prev:
                continue
            occupant = max(prev)
            last_read = max(e for v in waves for e, x in v.reads if x == occupant)
            assert e_issue > last_read, f"WAR: {what} issued into buffer {b} in epoch {e_issue}, {occupant} still read in {last_read}"


@pytest.mark.parametrize("slabs", [4, 5, 7, 16, 33])
@pytest.mark.parametrize("lockstep", [False, True])
def test_pingpong_ring_of_four(slabs, lockstep):
    """mfma_f16_pp_kernel / mfma_i8_pp_kernel / mfma_f32_pp_kernel: 4-slab ring, 4 pieces per wave per
    slab, vmcnt(8) at the end of every load segment, groups one barrier apart."""
    waves = []
    for group in (0, 1):
        w = Wave()
        for s in range(3):
            w.issue(s, 4)
        w.wait_vmcnt(8)
        w.barrier()
        if group == 1 and not lockstep:
            w.barrier()
        for u in range(slabs):
            w.read(u)
            w.issue(u + 3, 4)            # past the end: harmless re-fetch into a dead buffer
            w.wait_vmcnt(8)
            w.barrier()                  # load segment ends; compute segment
            w.barrier()
        waves.append(w)
    check(waves, lambda slab: slab % 4)
    # both groups execute the same number of barriers once group 0 adds its final extra one
    assert waves[1].epoch - waves[0].epoch == (0 if lockstep else 1)


@pytest.mark.parametrize("slabs", [8, 10, 18, 32])
def test_pingpong_full_line_a_requests(slabs):
    """mfma_f16_pp2_kernel / mfma_i8_pp2_kernel: A in double slabs (ring of 3, two halves of 2 pieces
    per wave), B in slabs (ring of 4, 2 pieces per wave); prologue in the order of four virtual
    segments; vmcnt(8) at the end of every load segment."""
    A = lambda d: ("A", d)
    B = lambda s: ("B", s)
    waves = []
    for group in (0, 1):
        w = Wave()
        w.issue(("Ah", 0, 0), 2)
        w.issue(("Ah", 0, 1), 2)
        w.issue(B(0), 2)
        w.issue(("Ah", 1, 0), 2)
        w.issue(B(1), 2)
        w.issue(("Ah", 1, 1), 2)
        w.issue(B(2), 2)
        w.wait_vmcnt(8)
        w.barrier()
        if group == 1:
            w.barrier()
        for u in range(slabs):
            w.read(A(u // 2))
            w.read(B(u))
            w.issue(("Ah", u // 2 + 2, u % 2), 2)
            w.issue(B(u + 3), 2)
            w.wait_vmcnt(8)
            w.barrier()
            w.barrier()
        waves.append(w)
    # fold the two halves of an A double slab into one object: retired when both halves are, issued at the first
    for w in waves:
        for d in range(slabs // 2 + 2):
            halves = [w.retired.get(("Ah", d, h)) for h in (0, 1)]
            if all(h is not None for h in halves):
                w.retired[A(d)] = max(halves)
        w.issues = [(e, A(x[1]) if x[0] == "Ah" else x) for e, x in w.issues]

    def buffer_of(x):
        return ("A", x[1] % 3) if x[0] == "A" else ("B", x[1] % 4)
    check(waves, buffer_of)


@pytest.mark.parametrize("slabs", [1, 2, 3, 7, 12, 31])
def test_split_kernel_one_barrier_per_stage(slabs):
    """mfma_f32_split_kernel, default schedule: ring of 3 stages, 6 pieces per wave per stage, three stages issued in
    the prologue; per stage: reads of this stage, vmcnt(6), barrier, refill of this stage's buffer with stage s+3,
    reads of the first fragments of stage s+1."""
    waves = []
    for _ in range(8):
        w = Wave()
        for s in range(3):
            w.issue(s, 6)
        w.wait_vmcnt(12)
        w.barrier()
        w.read(0)
        for s in range(slabs):
            w.read(s)
            w.wait_vmcnt(6)
            w.barrier()
            w.issue(s + 3, 6)
            w.read(s + 1)              # B fragments and the first A fragments of the next stage
        waves.append(w)
    check(waves, lambda st: st % 3)


@pytest.mark.parametrize("slabs", [1, 2, 3, 7, 12, 31])
def test_split_kernel_pingpong_ring_of_three(slabs):
    """mfma_f32_split_kernel, ping-pong schedule: two stages issued in the prologue; per stage a load segment (reads
    of stage s, DMA of stage s+2 into the buffer of stage s-1, vmcnt(6)), barrier, compute segment, barrier; group 1
    one barrier behind group 0."""
    waves = []
    for group in (0, 1):
        w = Wave()
        w.issue(0, 6)
        w.issue(1, 6)
        w.wait_vmcnt(6)
        w.barrier()
        if group == 1:
            w.barrier()
        for s in range(slabs):
            w.read(s)
            w.issue(s + 2, 6)
            w.wait_vmcnt(6)
            w.barrier()
            w.barrier()
        waves.append(w)
    check(waves, lambda st: st % 3)
    assert waves[1].epoch - waves[0].epoch == 1


@pytest.mark.parametrize("slabs", [1, 2, 3, 9])
def test_valu_tile_dma_double_buffer(slabs):
    """valu_tile_dma_kernel: ring of 2, per slab: vmcnt(0), barrier, issue the next slab, compute this one."""
    waves = []
    for _ in range(4):
        w = Wave()
        w.issue(0, 4)
        for t in range(slabs):
            w.wait_vmcnt(0)
            w.barrier()
            if t + 1 < slabs:
                w.issue(t + 1, 4)
            w.read(t)
        waves.append(w)
    check(waves, lambda slab: slab % 2)


def test_the_model_catches_a_wrong_count():
    """The same ring-of-four program with vmcnt(12) (one slab too lenient) must violate RAW, and a
    refill issued one segment early must violate WAR -- otherwise the checks above prove nothing."""
    waves = []
    for group in (0, 1):
        w = Wave()
        for s in range(3):
            w.issue(s, 4)
        w.wait_vmcnt(8)
        w.barrier()
        if group == 1:
            w.barrier()
        for u in range(8):
            w.read(u)
            w.issue(u + 3, 4)
            w.wait_vmcnt(12)
            w.barrier()
            w.barrier()
        waves.append(w)
    with pytest.raises(AssertionError, match="RAW"):
        check(waves, lambda slab: slab % 4)
    waves = []
    for group in (0, 1):
        w = Wave()
        for s in range(3):
            w.issue(s, 4)
        w.wait_vmcnt(8)
        w.barrier()
        if group == 1:
            w.barrier()
        for u in range(8):
            w.read(u)
            w.issue(u + 4, 4)            # into the buffer the OTHER group is still reading
            w.wait_vmcnt(8)
            w.barrier()
            w.barrier()
        waves.append(w)
    with pytest.raises(AssertionError):
        check(waves, lambda slab: slab % 4)


# ---------------------------------------------------------------------------------------------------------------------
# Stream-K unit ranges (gemm_hls_amd/csrc/mm_mfma_f32_streamk.inc: mfma_f32_streamk_kernel / streamk_fixup_kernel), replayed on
# the CPU with the kernels' own integer formulas: every (tile, slab) unit is multiplied exactly once, a workgroup writes at
# most one head and one tail slot, and the fix-up of a tile adds exactly the slots that were written for it, in ascending k.
def _sk_begin(units, w, nwg):
    return units * w // nwg


def _streamk_main(tiles, spt, nwg):
    """What the main kernel does: returns (direct: {tile}, slots: {(w, which): (tile, s0, s1)})."""
    units = tiles * spt
    direct, slots, covered = set(), {}, {}
    for w in range(nwg):
        u0, u1 = _sk_begin(units, w, nwg), _sk_begin(units, w + 1, nwg)
        first_tile = u0 // spt
        u = u0
        while u < u1:
            tile = u // spt
            s0 = u - tile * spt
            s1 = min(spt, s0 + (u1 - u))
            for sl in range(s0, s1):
                assert (tile, sl) not in covered
                covered[(tile, sl)] = w
            if s0 == 0 and s1 == spt:
                assert tile not in direct
                direct.add(tile)
            else:
                key = (w, int(tile != first_tile))
                assert key not in slots, "a workgroup reused a scratch slot"
                slots[key] = (tile, s0, s1)
            u += s1 - s0
    assert len(covered) == units
    return direct, slots


def _streamk_fixup(tile, tiles, spt, nwg):
    """What the fix-up workgroup of `tile` reads: None if one range holds the whole tile, else the slot keys in order."""
    units = tiles * spt
    u_lo, u_hi = tile * spt, tile * spt + spt
    w = u_lo * nwg // units
    while w + 1 < nwg and _sk_begin(units, w + 1, nwg) <= u_lo:
        w += 1
    while w > 0 and _sk_begin(units, w, nwg) > u_lo:
        w -= 1
    if _sk_begin(units, w + 1, nwg) >= u_hi:
        return None
    keys = []
    while w < nwg and _sk_begin(units, w, nwg) < u_hi:
        if _sk_begin(units, w + 1, nwg) != _sk_begin(units, w, nwg):   # an empty range (fewer units than workgroups) wrote nothing
            first_tile = _sk_begin(units, w, nwg) // spt
            keys.append((w, int(tile != first_tile)))
        w += 1
    return keys


@pytest.mark.parametrize("tiles,spt,nwg", [(324, 72, 512), (400, 80, 512), (576, 96, 512), (784, 8, 512), (64, 32, 512), (1, 3, 512),
                                           (513, 1, 512), (1000, 7, 512), (7, 1000, 512), (257, 33, 512), (1024, 128, 512), (3, 2, 8)])
def test_streamk_ranges_cover_every_unit_once_and_fixup_reads_what_was_written(tiles, spt, nwg):
    direct, slots = _streamk_main(tiles, spt, nwg)
    by_tile = {}
    for key, (tile, s0, s1) in slots.items():
        by_tile.setdefault(tile, []).append((s0, s1, key))
    for tile in range(tiles):
        keys = _streamk_fixup(tile, tiles, spt, nwg)
        if tile in direct:
            assert keys is None and tile not in by_tile      # written to C by the main kernel, nothing to add
            continue
        segs = sorted(by_tile[tile])
        assert keys == [k for _, _, k in segs]               # exactly the written slots, in ascending k
        assert segs[0][0] == 0 and segs[-1][1] == spt and all(a[1] == b[0] for a, b in zip(segs, segs[1:]))


def test_streamk_ranges_random_configurations():
    import random
    rnd = random.Random(7)
    for _ in range(300):
        tiles, spt, nwg = rnd.randint(1, 1500), rnd.randint(1, 200), rnd.choice([8, 64, 256, 512])
        direct, slots = _streamk_main(tiles, spt, nwg)
        split = {t for (t, _, _) in slots.values()}
        assert not (split & direct) and len(split | direct) == tiles
        for tile in rnd.sample(sorted(split), min(20, len(split))):
            keys = _streamk_fixup(tile, tiles, spt, nwg)
            assert keys == [k for _, _, k in sorted((s0, s1, k) for k, (t, s0, s1) in slots.items() if t == tile)]


# ---------------------------------------------------------------------------------------------------------------------
# Stream-K in teams (mfma_f32_streamk_teams_kernel: the last arriver gathers, or a fix-up kernel does), replayed
# with the kernel's integer formulas: teams of sr x sc workgroups walk equal ranges of (super-tile, slab) units; a segment
# that does not begin a tile is its workgroup's FIRST one (-> slot + flag), a segment that begins a tile but does not end it
# is its workgroup's LAST one (-> C, then waits for and adds the slots of the following teams' same lane).  Checked: every
# (tile, slab) multiplied once; the gather of a cut tile adds exactly the slots written for it, in ascending k; every wait
# aims at a flag that is raised by a segment which itself never waits (no cycle), and no slot is written twice.
def _team_side(t):
    return 4 if t % 4 == 0 else 2 if t % 2 == 0 else 1


def _ordered_replay(tiles_n, tiles_m, spt):
    sr, sc = _team_side(tiles_n), _team_side(tiles_m)
    lanes, tpx = sr * sc, 64 // (sr * sc)
    st_rows, st_cols = -(-tiles_n // sr), -(-tiles_m // sc)
    units = st_rows * st_cols * spt
    teams = max(1, min(8 * tpx, units // 8, 8 * st_rows * st_cols))
    covered, slot_of, waits, whole, lowest = {}, {}, [], set(), {}
    for block in range(512):
        xcd, place = block % 8, block // 8
        team_in_xcd, lane = place // lanes, place % lanes
        team = team_in_xcd * 8 + xcd
        if team_in_xcd >= tpx or team >= teams:
            continue
        w = team * lanes + lane
        u0, u1 = _sk_begin(units, team, teams), _sk_begin(units, team + 1, teams)
        assert u1 > u0
        u, index = u0, 0
        segments = []
        while u < u1:
            st = u // spt
            s0 = u - st * spt
            s1 = min(spt, s0 + (u1 - u))
            u += s1 - s0
            tile = ((st % st_rows) * sr + lane % sr, (st // st_rows) * sc + lane // sr)
            assert tile[0] < tiles_n and tile[1] < tiles_m          # the team shape divides the grid
            segments.append((tile, s0, s1))
        for index, (tile, s0, s1) in enumerate(segments):
            for sl in range(s0, s1):
                assert (tile, sl) not in covered
                covered[(tile, sl)] = w
            if s0 == 0 and s1 == spt:
                whole.add(tile)
            elif s0 > 0:
                assert index == 0 and w not in slot_of              # first segment; one slot per workgroup
                slot_of[w] = (tile, s0, s1)
            else:
                assert index == len(segments) - 1                   # the waiting segment is the last thing it does
                st = (tile[0] // sr) + (tile[1] // sc) * st_rows
                u_hi = st * spt + spt
                t_end = team + 1
                while t_end < teams and _sk_begin(units, t_end, teams) < u_hi:
                    t_end += 1
                lowest[tile] = (s1, [o * lanes + lane for o in range(team + 1, t_end)])
    assert len(covered) == tiles_n * tiles_m * spt
    for tile, (s1, sources) in lowest.items():
        assert 1 <= len(sources) <= 8, "a cut tile gathers from 1..8 following teams"
        k = s1
        for o in sources:                                           # ascending k, contiguous, each one really written for this tile
            t, a, b = slot_of[o]
            assert t == tile and a == k
            k = b
        assert k == spt
    cut = {t for (t, _, _) in slot_of.values()}
    assert cut == set(lowest) and not (cut & whole) and len(cut | whole) == tiles_n * tiles_m
    # The two-kernel form (HANDOVER = false, what MM_PATH_AUTO runs) does the gather in streamk_teams_fixup_kernel: one
    # workgroup per tile re-derives, from the tile alone, which slots to add.  It must find exactly what the hand-over
    # form's waiting workgroup gathered: nothing for a whole tile, the same slots in the same order for a cut one.
    for tile_c in range(tiles_m):
        for tile_r in range(tiles_n):
            st = (tile_c // sc) * st_rows + tile_r // sr
            lane = tile_r % sr + (tile_c % sc) * sr
            u_lo, u_hi = st * spt, st * spt + spt
            t0 = u_lo * teams // units
            while t0 + 1 < teams and _sk_begin(units, t0 + 1, teams) <= u_lo:
                t0 += 1
            while t0 > 0 and _sk_begin(units, t0, teams) > u_lo:
                t0 -= 1
            t_end = t0 + 1
            while t_end < teams and _sk_begin(units, t_end, teams) < u_hi:
                t_end += 1
            sources = [o * lanes + lane for o in range(t0 + 1, t_end)]
            if (tile_r, tile_c) in whole:
                assert sources == []
            else:
                assert sources == lowest[(tile_r, tile_c)][1]
                assert covered[((tile_r, tile_c), 0)] == t0 * lanes + lane       # C holds the part of the team the fix-up starts from
    # The last-arriver form (Combine::LastArriver, what MM_PATH_AUTO runs): EVERY part of a cut tile -- whichever it is --
    # derives the tile's part list (t0 .. t_end - 1) from its own team and segment, and the slot / flag index of each part
    # (2w for a first segment, 2w + 1 for the lowest-k part = its owner's last segment).  All parts of a tile must derive
    # the same list, the indices must be distinct across the launch, and the list must be the hand-over form's gather list.
    used = {}
    for block in range(512):
        xcd, place = block % 8, block // 8
        team_in_xcd, lane = place // lanes, place % lanes
        team = team_in_xcd * 8 + xcd
        if team_in_xcd >= tpx or team >= teams:
            continue
        u0, u1 = _sk_begin(units, team, teams), _sk_begin(units, team + 1, teams)
        u = u0
        while u < u1:
            st = u // spt
            s0 = u - st * spt
            s1 = min(spt, s0 + (u1 - u))
            u += s1 - s0
            if s0 == 0 and s1 == spt:
                continue
            tile = ((st % st_rows) * sr + lane % sr, (st // st_rows) * sc + lane // sr)
            u_lo, u_hi = st * spt, st * spt + spt
            t0 = team
            while t0 > 0 and _sk_begin(units, t0, teams) > u_lo:
                t0 -= 1
            t_end = team + 1
            while t_end < teams and _sk_begin(units, t_end, teams) < u_hi:
                t_end += 1
            index = lambda o: 2 * (o * lanes + lane) + (1 if o == t0 else 0)
            assert (s0 == 0) == (team == t0)                                     # the lowest-k part is t0's, and only t0's
            assert [o * lanes + lane for o in range(t0 + 1, t_end)] == lowest[tile][1]
            mine = index(team)
            assert mine not in used and mine < 2 * 512                           # one writer per slot / flag in the launch
            used[mine] = tile
            for o in range(t0, t_end):                                           # every sibling it looks at really is a part of this tile
                k_lo = max(_sk_begin(units, o, teams), u_lo) - u_lo
                assert covered[(tile, k_lo)] == o * lanes + lane
    assert len(used) == len(slot_of) + len(lowest)
    return sr, sc, teams


@pytest.mark.parametrize("tiles_n,tiles_m,spt", [(18, 18, 72), (20, 20, 80), (24, 24, 96), (28, 28, 8), (31, 31, 124), (40, 40, 160), (1, 1, 3),
                                                 (1, 1, 1024), (8, 8, 32), (3, 24, 33), (19, 18, 72), (32, 32, 128), (2, 2, 2), (24, 5, 9), (4, 4, 1)])
def test_streamk_ordered_hand_over_replayed(tiles_n, tiles_m, spt):
    sr, sc, teams = _ordered_replay(tiles_n, tiles_m, spt)
    assert tiles_n % sr == 0 and tiles_m % sc == 0 and 1 <= teams <= 512


def test_streamk_ordered_hand_over_random_grids():
    import random
    rnd = random.Random(11)
    for _ in range(150):
        _ordered_replay(rnd.randint(1, 45), rnd.randint(1, 45), rnd.randint(1, 160))


# ---------------------------------------------------------------------------------------------------------------------
# The 64 x 64 fp32 geometry (mfma_f32_small_kernel): ring of four stages, four DMA pieces per wavefront per slab.  Prologue:
# slabs 0..2 whole, wait vmcnt(8), barrier.  During slab t the pieces of slab t+3 are issued one per k-group (the fourth
# right before the hand-over), then vmcnt(8) + barrier, then slab t+1 is read.  One workgroup, four wavefronts in step.
@pytest.mark.parametrize("slabs", [1, 2, 3, 4, 5, 9, 32, 129])
def test_f32_small_geometry_ring_of_four(slabs):
    waves = []
    for _ in range(4):
        w = Wave()
        for s in range(3):
            w.issue(s, 4)
        w.wait_vmcnt(8)
        w.barrier()
        w.read(0)                         # first fragments of slab 0 (read_group(0, 0, f0))
        for t in range(slabs - 1):        # the steady slabs; the last slab is read in the tail loop without refills
            for g in range(4):            # piece g of slab t+3 after the first MFMA of group g, into the stage slab t-1 left
                w.read(t)
                w.issue((t + 3, g), 1)
            w.wait_vmcnt(8)               # everything but the 8 newest pieces: slab t+1 has landed
            w.barrier()
            w.read(t + 1)                 # first fragments of slab t+1
        w.read(slabs - 1)
        waves.append(w)
    for w in waves:                       # fold the four single pieces of a slab into one object
        for s in range(3, slabs + 2):
            got = [w.retired.get((s, g)) for g in range(4)]
            if all(x is not None for x in got):
                w.retired[s] = max(got)
        first = {}
        for e, what in w.issues:
            if isinstance(what, tuple):
                first.setdefault(what[0], e)
        w.issues = [(e, x) for e, x in w.issues if not isinstance(x, tuple)] + [(e, s) for s, e in first.items()]
    check(waves, lambda slab: slab % 4)


def test_f32_small_geometry_a_shorter_ring_would_be_caught():
    """The same program on a ring of three stages refills the stage that is being read: the checker must object."""
    waves = []
    for _ in range(4):
        w = Wave()
        for s in range(2):
            w.issue(s, 4)
        w.wait_vmcnt(4)
        w.barrier()
        w.read(0)
        for t in range(6):
            w.read(t)
            w.issue(t + 2, 4)             # into stage (t + 2) % 3 == (t - 1) % 3 ... but declared as a ring of TWO below
            w.wait_vmcnt(4)
            w.barrier()
            w.read(t + 1)
        waves.append(w)
    with pytest.raises(AssertionError):
        check(waves, lambda slab: slab % 2)
