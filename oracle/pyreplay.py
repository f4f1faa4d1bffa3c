"""TEST INFRASTRUCTURE — Python restatement of the Trainer's replay handling (train.py:156-201 remove_duplicates,
226-236 FIFO trim), pinned to tests/golden/replay.json (produced by running the reference's Trainer methods).
Pure-Python loops: used on small cases only."""


def remove_duplicates(flat):
    """train.py:172-198.  `flat` = list of [key, board, pi list, z]; the FIRST occurrence of every key is mutated
    in place (the reference keeps a reference to that list object) and returned, in first-occurrence order."""
    by_key, counts, pol_counts = {}, {}, {}
    for item in flat:
        k = item[0]
        if k in by_key:
            first = by_key[k]
            if item[2] and first[2]:
                first[2] = [sum(x) for x in zip(first[2], item[2])]
                pol_counts[k] += 1
            elif item[2]:
                first[2] = item[2]
            first[3] += item[3]
            counts[k] += 1
        else:
            by_key[k] = item
            counts[k] = 1
            pol_counts[k] = 1
    for k, first in by_key.items():
        if first[2]:
            first[2] = [x / pol_counts[k] for x in first[2]]
        first[3] = first[3] / counts[k]
    return list(by_key.values())


def fifo_append(buffer, games, n_games_buffer):
    """train.py:226-236."""
    for g in games:
        buffer.append(g)
    while len(buffer) > n_games_buffer:
        del buffer[0]
    return buffer
