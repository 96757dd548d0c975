"""What would scheduling the blocks of the wave loop by demand buy?  A Monte-Carlo model of one wave.

The trace kernel runs three blocks one after the other, each with the lanes that need it: the search
(+ absorption check), the element operation, and the renew block (tallies of ended photons, pathway
and re-emission of absorbed ones, new photon ids).  A block costs its wave cycles whatever the number
of lanes in it.  The model holds 64 lanes (optionally K photons per lane), moves photons between the
states S (needs a search), O (needs an element operation), R (needs the renew block) with the
transition probabilities of a configuration's tallies, charges every executed block its cost as
measured by the cycle stamps of the diagnostic build (profiles/*_sections.txt), and compares
schedules by wave cycles per photon:

  current    search for all S lanes, the operation at once for those not absorbed, the renew block
             when the lane-iterations spent waiting reach a threshold (the kernel's rule)
  demand     three-phase: run the block with the most waiting lanes (times a weight per block)
  K = 2, 3   several photons per lane, the block run on whichever of a lane's photons needs it
             (with and without the cost of swapping a parked photon's registers in)

usage: python tools/phase_sim.py [photons]      (CPU only)
"""
import random
import sys

# C2 (profiles/r02_sections.txt): wave cycles per pass of the block, independent of the lanes in it
C_SEARCH, C_OP, C_RENEW = 3469.0 + 1187.0, 2940.0, 6845.0
# C2 tallies: 6.33 iterations per photon, 30 % of the searches end in an absorption, 5 % of the
# absorptions in internal conversion, a fifth of the operations on a detector
P_ABSORB, P_CONVERT, P_DETECT = 0.30, 0.05, 0.204
S, O, R = 0, 1, 2
COST = (C_SEARCH, C_OP, C_RENEW)


def advance(state):
    """(new state, photons ended)"""
    if state == S:
        if random.random() < P_ABSORB:
            return R, (1 if random.random() < P_CONVERT else 0)
        return O, 0
    if state == O:
        return (R, 1) if random.random() < P_DETECT else (S, 0)
    return S, 0


def current(n_photons, debt_limit):
    lanes, done, t, debt = [R] * 64, 0, 0.0, 0
    passes, filled = [0, 0, 0], [0, 0, 0]
    while done < n_photons:
        waiting = lanes.count(R)
        debt += waiting
        if waiting and (debt >= debt_limit or lanes.count(S) == 0):
            debt = 0
            t += C_RENEW; passes[R] += 1; filled[R] += waiting
            lanes = [S if s == R else s for s in lanes]
        n_s = lanes.count(S)
        if n_s:
            t += C_SEARCH; passes[S] += 1; filled[S] += n_s
            n_o = 0
            for i, s in enumerate(lanes):
                if s != S:
                    continue
                s, d = advance(S)
                done += d
                if s == O:
                    n_o += 1
                    s, d = advance(O)
                    done += d
                lanes[i] = s
            if n_o:
                t += C_OP; passes[O] += 1; filled[O] += n_o
    return t / n_photons, [filled[k] / max(1, passes[k]) for k in range(3)]


def demand(n_photons, photons_per_lane=1, weights=(1.0, 1.0, 1.0), swap_cost=0.0):
    lanes, done, t = [[R] * photons_per_lane for _ in range(64)], 0, 0.0
    passes, filled = [0, 0, 0], [0, 0, 0]
    while done < n_photons:
        ready = [sum(1 for lane in lanes if x in lane) for x in (S, O, R)]
        block = max(range(3), key=lambda k: ready[k] * weights[k])
        t += COST[block] + (swap_cost if photons_per_lane > 1 else 0.0)
        passes[block] += 1; filled[block] += ready[block]
        for lane in lanes:
            if block in lane:
                i = lane.index(block)
                lane[i], d = advance(block)
                done += d
    return t / n_photons, [filled[k] / max(1, passes[k]) for k in range(3)]


def show(label, result):
    cycles, lanes = result
    print("%-58s %6.0f wave cycles per photon; lanes in search / operation / renew %4.1f %4.1f %4.1f"
          % (label, cycles, lanes[S], lanes[O], lanes[R]))


if __name__ == "__main__":
    n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 30000
    random.seed(1)
    ideal = (6.33 * C_SEARCH + 6.33 * (1 - P_ABSORB) * C_OP + 2.9 * C_RENEW) / 64.0
    print("every block always with 64 lanes (bound): %.0f wave cycles per photon" % ideal)
    for limit in (24, 32, 40, 64):
        show("current schedule, renew threshold %d" % limit, current(n, limit))
    for w in ((1, 1, 1), (1, 1, 0.7), (1, 1.5, 0.6), (0.8, 1, 0.5), (1, 2, 0.7), (1, 1.5, 0.8)):
        show("demand schedule, weights %s" % (w,), demand(n, 1, w))
    for k in (2, 3):
        show("%d photons per lane, demand schedule" % k, demand(n, k))
        show("%d photons per lane, + 20 register swaps per block (340 cycles)" % k, demand(n, k, swap_cost=340.0))
