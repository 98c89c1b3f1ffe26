"""A model of the slice parse's pass (round 5, profiles/r05_parse_notes.md): 4096 resident wavefronts draw batches of 64
slices by ticket, longest first; a wavefront's batch takes `turns` turns, a turn costs `cost` vector instructions, and the
four wavefronts of a SIMD share its issue port (one instruction every 4 clocks).  Reproduces the measured pass (2.81 ms
modelled, 2.83 measured, 82 % of the issue slots filled) and prices the alternatives: what the intra wavefronts' turns,
the predicted wavefronts' instructions per turn and the end-of-pass tail are each worth.
    python tools/parse_schedule_model.py"""
import heapq
import random

random.seed(1)
GHZ = 2.29e9           # GRBM_GUI_ACTIVE / 8 XCDs / the pass's time (profiles/r05c_parse_counters.txt)


def sim(jobs, cost, slots_per_simd=4, n_simd=1024, lmin=2600):
    """jobs: [(turns, kind)] in ticket order; cost[kind]: vector instructions per turn.  A SIMD's wavefronts take their turns
    round robin: a round costs max(4 x the sum of their costs, lmin) clocks (lmin: a lone wavefront's latency-bound turn)."""
    nxt, simds = 0, []
    for _ in range(n_simd):
        waves = []
        for _ in range(slots_per_simd):
            if nxt < len(jobs):
                waves.append(list(jobs[nxt]))
                nxt += 1
        simds.append(waves)
    heap = [(0.0, s) for s in range(n_simd)]
    heapq.heapify(heap)
    end, busy, ends, last_ticket = 0.0, 0.0, [0.0] * n_simd, None
    while heap:
        tt, s = heapq.heappop(heap)
        waves = simds[s]
        if not waves:
            continue
        r = min(w[0] for w in waves)
        issue = 4 * sum(cost[w[1]] for w in waves)
        busy += r * issue
        tt += r * max(issue, lmin)
        new = []
        for w in waves:
            w[0] -= r
            if w[0] > 0:
                new.append(w)
            elif nxt < len(jobs):
                new.append(list(jobs[nxt]))
                nxt += 1
                if nxt == len(jobs):
                    last_ticket = tt
        simds[s] = new
        ends[s] = tt
        end = max(end, tt)
        if new:
            heapq.heappush(heap, (tt, s))
    return end, busy / (end * n_simd), sorted(ends), last_ticket


# cfg2: 680 batches of intra slices (1090 +- 15 turns), 7480 of predicted ones (372 .. 650 turns, mean 478): tools/parse_stats.py
INTRA = [(int(random.gauss(1090, 15)), "I") for _ in range(680)]
PRED = sorted([(int(372 + (650 - 372) * (random.random() ** 1.6)), "P") for _ in range(7480)], reverse=True)


def scaled(jobs, f):
    return [(int(t * f), k) for t, k in jobs]


if __name__ == "__main__":
    rows = [("round-5 kernel as measured (289 / 316 instructions per turn)", INTRA + PRED, {"I": 289, "P": 316}),
            ("intra wavefronts: DC fused (0.72 x the turns, 340 per turn)", scaled(INTRA, 0.72) + PRED, {"I": 340, "P": 316}),
            ("predicted wavefronts: 10 % fewer instructions per turn", INTRA + PRED, {"I": 289, "P": 285}),
            ("both", scaled(INTRA, 0.72) + PRED, {"I": 340, "P": 285}),
            ("both, predicted at 270", scaled(INTRA, 0.72) + PRED, {"I": 340, "P": 270})]
    for label, jobs, cost in rows:
        end, busy, ends, last = sim(jobs, cost)
        print("%-66s %.2f ms, issue slots filled %.2f, last ticket drawn at %.2f ms, SIMDs through at %.2f / %.2f / %.2f ms (10 / 50 / 90 %%)"
              % (label, end / GHZ * 1e3, busy, last / GHZ * 1e3, ends[102] / GHZ * 1e3, ends[512] / GHZ * 1e3, ends[921] / GHZ * 1e3))
