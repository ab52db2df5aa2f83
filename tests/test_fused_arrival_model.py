"""CPU model of the arrival counter that lets the overlapped launch number itself (csrc/sa_fused.hip: fused_arrive,
PN2_GENERATION_DEVICE). The device code is two atomic adds on one 64-bit word per cloud; the model replays them in random
interleavings and checks the two properties the hand-off rests on: every participant of a launch derives the SAME tag, and
consecutive launches on a workspace derive DIFFERENT, non-zero tags -- for any number of participants per launch, with the last
arriver's second add landing anywhere before the launch ends, and across the 32-bit wrap of the tag.
(What the reference has in this place is nothing: pointnet_util.py:40-46 calls the operators one after the other.)"""
import random

MASK = (1 << 64) - 1


def _arrive(word, npc):
    """One workgroup's fused_arrive on `word` (a one-element list): returns (tag, second add or None)."""
    old = word[0]
    word[0] = (word[0] + 1) & MASK
    late = (65536 - npc) if (old & 0xffff) + 1 == npc else None
    return (old >> 16) % 0xffffffff + 1, late


def _launch(word, npc, rng):
    tags, pending = [], None
    for i in range(npc):
        tag, late = _arrive(word, npc)
        tags.append(tag)
        if late is not None:
            assert pending is None and i == npc - 1            # exactly one last arriver, and it is the last
            pending = late
    # the second add is performed some time before the launch ends (a launch ends when its memory operations have)
    word[0] = (word[0] + pending) & MASK
    return tags


def test_all_participants_agree_and_consecutive_launches_differ():
    rng = random.Random(0)
    for start in (0, 5 << 16, (0xffffffff - 3) << 16, ((1 << 48) - 7) << 16):
        word = [start]
        prev = None
        for launch in range(2000):
            npc = rng.choice((2, 2, 2, 3, 5, 9, 17, 129))
            tags = _launch(word, npc, rng)
            assert len(set(tags)) == 1 and tags[0] != 0 and tags[0] <= 0xffffffff
            assert tags[0] != prev
            prev = tags[0]
            assert word[0] & 0xffff == 0                       # arrivals back to zero for the next launch
        assert (word[0] >> 16) == ((start >> 16) + 2000) & ((1 << 48) - 1)


def test_counters_of_different_clouds_are_independent():
    rng = random.Random(1)
    words = [[0] for _ in range(8)]
    for launch in range(300):
        npc = rng.choice((2, 3, 5))
        order = [c for c in range(8) for _ in range(npc)]
        rng.shuffle(order)                                      # workgroups of all clouds arrive in any order
        tags = {c: [] for c in range(8)}
        late = {}
        for c in order:
            t, l = _arrive(words[c], npc)
            tags[c].append(t)
            if l is not None:
                late[c] = l
        for c, l in late.items():
            words[c][0] = (words[c][0] + l) & MASK
        assert all(set(tags[c]) == {launch + 1} for c in range(8))
