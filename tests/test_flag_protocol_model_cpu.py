"""A model of the peer-mapped transports' ordering protocol (hodor_amd/csrc/abi_exchange.hip, "direct transport"; the
schedule on top of it in abi_dist.hip), explored exhaustively over every interleaving of the ranks' in-order streams:

    producer:  begin   wait released[t] >= g - 1 for all t (in MY flag block)   the slot may be overwritten
               write   my slab of generation g into every peer's receive buffer of the slot
               signal  arrived[me] := g in every peer's flag block
    consumer:  wait    arrived[s] >= g for all s (in MY flag block)
               read    my receive buffer of the slot
               release released[me] := g in every peer's flag block

No box of the pool has two GPUs, so the protocol has only ever run between processes that share one device; this model is
the part of "correct by construction" that can be checked without hardware: no slab is overwritten before its reader has
read it, no reader sees a slab of the wrong generation, and no schedule the library accepts can deadlock — including the
split-phase pair (two transforms in flight on one stream), which DOES deadlock on a handle with a single slot: the library
refuses that case (hodor_dist_ntt_begin_dev: "a handle with N slots carries at most N transforms in flight")."""
import pytest


class _Slots:
    """abi_dist.hip's claim rule (round 6): the LOWEST slot no open operation holds, given back when the operation's
    release has been enqueued.  `round_robin` = the rule of rounds 4-5, kept to show the sequence it breaks on."""

    def __init__(self, n_slots, round_robin=False):
        self.busy, self.gen, self.nxt, self.rr = [False] * n_slots, [0] * n_slots, 0, round_robin

    def claim(self):
        if self.rr:
            s = self.nxt
            self.nxt = (self.nxt + 1) % len(self.busy)
        else:
            free = [i for i, b in enumerate(self.busy) if not b]
            if not free:
                raise RuntimeError("refused: every slot is held by an open operation")    # HODOR_ERR_INVALID
            s = free[0]
        self.busy[s] = True
        self.gen[s] += 1
        return s, self.gen[s]

    def give_back(self, s):
        self.busy[s] = False


def _program(n_slots, rounds, in_flight, single_slot_unchecked=False):
    """One rank's stream: `rounds` groups of `in_flight` transforms, all begins of a group before its ends (the split-phase
    pair of abi_dist.hip: begin A, begin B, end A, end B)."""
    ops = []
    slots = _Slots(n_slots, round_robin=single_slot_unchecked)
    for _ in range(rounds):
        group = [slots.claim() for _ in range(in_flight)]
        for s, g in group:
            ops += [("begin", s, g), ("write", s, g), ("signal", s, g)]
        for s, g in group:
            ops += [("wait", s, g), ("read", s, g), ("release", s, g)]
            slots.give_back(s)
    return ops


def _program_ends_out_of_order(n_slots, round_robin):
    """begin A, begin B, end B, begin C, end A, end C — ends are not FIFO (the header never asked for that)."""
    slots = _Slots(n_slots, round_robin)
    ops = []

    def begin():
        s, g = slots.claim()
        ops.extend([("begin", s, g), ("write", s, g), ("signal", s, g)])
        return s, g

    def end(sg):
        s, g = sg
        ops.extend([("wait", s, g), ("read", s, g), ("release", s, g)])
        slots.give_back(s)

    a, b = begin(), begin()
    end(b)
    c = begin()
    end(a)
    end(c)
    return ops


def _program_natural(n_slots, calls):
    """hodor_dist_ntt_natural_dev: an exchange whose slot is released only AFTER the transform that reads it — and that
    transform runs an exchange of its own — then a third exchange (abi_dist.hip)."""
    ops = []
    slots = _Slots(n_slots, round_robin=n_slots == 1)   # one slot: the sequence the library refuses, modelled unchecked

    for _ in range(calls):
        a, ga = slots.claim()
        ops += [("begin", a, ga), ("write", a, ga), ("signal", a, ga), ("wait", a, ga), ("read", a, ga)]
        t, gt = slots.claim()
        ops += [("begin", t, gt), ("write", t, gt), ("signal", t, gt), ("wait", t, gt), ("read", t, gt), ("release", t, gt)]
        slots.give_back(t)
        ops += [("release", a, ga)]
        slots.give_back(a)
        b, gb = slots.claim()
        ops += [("begin", b, gb), ("write", b, gb), ("signal", b, gb), ("wait", b, gb), ("read", b, gb), ("release", b, gb)]
        slots.give_back(b)
    return ops


def _explore(n_ranks, n_slots, rounds, in_flight, producer_waits=True, prog=None):
    """DFS over all interleavings.  Returns (number of states, deadlocked?).  Raises AssertionError on a data hazard."""
    prog = prog or _program(n_slots, rounds, in_flight)
    P, S = n_ranks, n_slots
    # state: pc per rank; arrived[r][s][from]; released[r][s][from]; recv[r][s][from] = generation of the slab lying there;
    # unread[r][s][from] = True between the write of a slab and its read
    zero = tuple(tuple(tuple(0 for _ in range(P)) for _ in range(S)) for _ in range(P))
    start = (tuple(0 for _ in range(P)), zero, zero, zero, zero)
    seen, stack, deadlock = {start}, [start], False

    def setcell(t, r, s, f, v):
        row = list(t[r][s])
        row[f] = v
        slot = list(t[r])
        slot[s] = tuple(row)
        out = list(t)
        out[r] = tuple(slot)
        return tuple(out)

    while stack:
        pcs, arrived, released, recv, unread = state = stack.pop()
        moved = False
        for r in range(P):
            if pcs[r] == len(prog):
                continue
            op, s, g = prog[pcs[r]]
            a2, rel2, rc2, un2 = arrived, released, recv, unread
            if op == "begin":
                if producer_waits and any(released[r][s][t] < g - 1 for t in range(P)):
                    continue                                    # blocked: a peer still reads what I sent last time
            elif op == "write":
                for t in range(P):
                    assert not unread[t][s][r], "rank %d overwrites a slab rank %d has not read (slot %d, gen %d)" % (r, t, s, g)
                    rc2 = setcell(rc2, t, s, r, g)
                    un2 = setcell(un2, t, s, r, 1)
            elif op == "signal":
                for t in range(P):
                    a2 = setcell(a2, t, s, r, g)
            elif op == "wait":
                if any(arrived[r][s][t] < g for t in range(P)):
                    continue                                    # blocked: a slab is still on its way
            elif op == "read":
                for t in range(P):
                    assert recv[r][s][t] == g, "rank %d reads generation %d from rank %d, expected %d" % (r, recv[r][s][t], t, g)
                    un2 = setcell(un2, r, s, t, 0)
            elif op == "release":
                for t in range(P):
                    rel2 = setcell(rel2, t, s, r, g)
            moved = True
            nxt = (pcs[:r] + (pcs[r] + 1,) + pcs[r + 1:], a2, rel2, rc2, un2)
            if nxt not in seen:
                seen.add(nxt)
                stack.append(nxt)
        if not moved and any(pc < len(prog) for pc in pcs):
            deadlock = True
    return len(seen), deadlock


@pytest.mark.parametrize("n_ranks,n_slots,rounds,in_flight", [
    (2, 1, 3, 1), (2, 2, 3, 1), (2, 2, 2, 2), (2, 3, 2, 2), (2, 4, 2, 2), (3, 1, 2, 1), (3, 2, 1, 2), (4, 1, 1, 1)])
def test_every_interleaving_is_safe_and_live(n_ranks, n_slots, rounds, in_flight):
    states, deadlock = _explore(n_ranks, n_slots, rounds, in_flight)
    assert states > 10 and not deadlock


def test_two_transforms_in_flight_on_one_slot_deadlock_which_is_why_the_library_refuses_them():
    """begin A, begin B on the same slot: B's begin waits for the release of A's generation, which A's end — behind it on
    the same in-order stream — would enqueue.  Every interleaving ends in that deadlock."""
    _, deadlock = _explore(2, 1, 1, 2, prog=_program(1, 1, 2, single_slot_unchecked=True))
    assert deadlock
    with pytest.raises(RuntimeError, match="refused"):      # ... and the claim rule itself refuses it
        _program(1, 1, 2)


def test_ends_in_any_order_are_live_with_lowest_free_first_and_were_not_with_round_robin():
    """Advisor, round 5: begin A (slot 0), begin B (slot 1), end B, begin C.  Round robin hands C slot 0 while A still holds
    it: C's begin waits for the release of A's generation, which A's end — later on the same stream — would enqueue.
    Lowest-free-first gives C the slot B gave back."""
    _, deadlock = _explore(2, 2, 0, 0, prog=_program_ends_out_of_order(2, round_robin=True))
    assert deadlock
    states, deadlock = _explore(2, 2, 0, 0, prog=_program_ends_out_of_order(2, round_robin=False))
    assert states > 10 and not deadlock
    states, deadlock = _explore(3, 2, 0, 0, prog=_program_ends_out_of_order(2, round_robin=False))
    assert not deadlock


@pytest.mark.parametrize("n_ranks,n_slots,calls", [(2, 2, 2), (2, 3, 1), (2, 4, 2), (3, 2, 1)])
def test_the_natural_order_schedule_is_safe_and_live_from_two_slots_on(n_ranks, n_slots, calls):
    states, deadlock = _explore(n_ranks, n_slots, 0, 0, prog=_program_natural(n_slots, calls))
    assert states > 10 and not deadlock


def test_the_natural_order_schedule_deadlocks_on_one_slot_which_is_why_the_library_refuses_it():
    _, deadlock = _explore(2, 1, 0, 0, prog=_program_natural(1, 1))
    assert deadlock


def test_the_begin_wait_is_what_prevents_the_overwrite():
    """The same exploration with the producer's wait removed finds the hazard the wait exists for: some interleaving
    stores generation 2 over a slab of generation 1 that its reader has not read."""
    with pytest.raises(AssertionError, match="overwrites a slab"):
        _explore(2, 1, 2, 1, producer_waits=False)
