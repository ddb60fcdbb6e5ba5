"""A second, deliberately different reading of the reference's tunnel builders, in plain Python (test infrastructure).

oracle/oracle_host.cpp restates Find_anchors::check_hits_order_conflict, ::define_tunnel and
::define_tunnel_with_overlapping_hits (src/utils/find_anchors.cpp:225-317, 320-447, 633-861) in C++; the product
(pagan2-msa_amd/csrc/host_anchors.cpp) is a third text.  This file reads the same source once more, statement by
statement and in its own shapes -- Python lists that grow with append / insert(0, .), the reference's variable names, the
loops in the reference's order -- so that the readings can be compared on random hit lists (parity is unpinned: the
reference cannot be built here, and independent readings that agree narrow what that leaves open).

A hit is (start_site_1, start_site_2, length, score): positions in the UNGAPPED child strings; str1 / str2 are the gapped
strings (skipped sites as '-').  Sorting: the reference sorts with std::sort and predicates that leave ties undefined; the
order-conflict reading below is therefore only compared on hit lists whose scores (and start sites) are distinct.
"""


def check_hits_order_conflict(len1, len2, hits, trim=5):
    """find_anchors.cpp:225-317.  hits: list of [s1, s2, length, score]; returns the surviving hits (new lists)."""
    hits = [list(h) for h in hits]
    hits.sort(key=lambda h: -h[3])                                   # sort_by_score: p.score > q.score          :230
    hit_site1 = [False] * len1                                       # :232-240
    hit_site2 = [False] * len2
    k = 0
    while k < len(hits):                                             # :246
        it1 = hits[k]
        # `it1->start_site_1+trim;` and the two `==` statements (:248-251) have no effect: only the length changes
        it1[2] -= trim * 2                                           # :252
        overlap = False
        i, j = it1[0], it1[1]
        while i < it1[0] + it1[2] and j < it1[1] + it1[2]:           # :255
            if hit_site1[i] or hit_site2[j]:
                overlap = True
                break
            i += 1
            j += 1
        if overlap:
            del hits[k]                                              # hits->erase(it1): the next hit moves into place
        else:
            i, j = it1[0], it1[1]
            while i < it1[0] + it1[2] and j < it1[1] + it1[2]:       # :271
                hit_site1[i] = True
                hit_site2[j] = True
                i += 1
                j += 1
            k += 1
    hits.sort(key=lambda h: (h[0], h[1]))                            # sort_by_start_site_1                       :280
    i1, i2 = 0, 1                                                    # it2 = it1 + 1                              :282-284
    while i1 < len(hits) and i2 < len(hits):                         # :285
        if hits[i1][1] > hits[i2][1]:                                # start_site_2 out of order                  :287
            if hits[i1][3] < hits[i2][3]:
                del hits[i1]                                         # erase(it1); it2 = it1 + 1
            else:
                del hits[i2]                                         # erase(it2); it2 = it1 + 1
            i2 = i1 + 1
            continue
        i1 += 1
        i2 += 1
    return hits


def define_tunnel(hits, str1, str2, width=15):
    """find_anchors.cpp:320-447.  Returns (upper_bound, lower_bound), each length1 + 1 entries."""
    length1, length2 = len(str1), len(str2)
    index1 = [i + 1 for i in range(length1) if str1[i] != '-']       # :329-339
    index2 = [i + 1 for i in range(length2) if str2[i] != '-']
    diagonals = [-1] * (length1 + 1)                                 # :345-348
    for (s1, s2, length, _score) in hits:                            # :350
        i = 0
        while i < length:                                            # :354
            diagonals[index1[s1 + i]] = index2[s2 + i]
            i += 1
        if s1 + i < len(index1) and index1[s1 + i] < len(diagonals): # :359
            diagonals[index1[s1 + i]] = -2
    upper_bound, lower_bound = [], []
    y1 = y2 = 0                                                      # :367-372
    prev_y = 0
    m_count = 0
    for i in range(0, length1 + 1):                                  # :374
        if i >= width and diagonals[i - width] >= 0:
            y1 = diagonals[i - width] + 0
        if diagonals[i] >= 0:
            y2 = diagonals[i] - width + 0
        if diagonals[i] >= 0 and i > 0 and diagonals[i - 1] + 1 == diagonals[i]:
            m_count += 1
        elif diagonals[i] == -2:
            m_count = 0
        y = min(y1, y2)
        y = max(y, 0)
        if diagonals[i] >= 0 and i > 0 and diagonals[i - 1] + 1 == diagonals[i] and m_count >= width:
            prev_y = y
        y = min(y, prev_y)
        y = max(y, 0)
        upper_bound.append(y)                                        # :405
    y1 = y2 = length2                                                # :409-413
    prev_y = length2
    m_count = 0
    for i in range(length1, -1, -1):                                 # :415
        if i <= length1 - width and diagonals[i + width] >= 0:
            y1 = diagonals[i + width] + 0
        if diagonals[i] >= 0:
            y2 = diagonals[i] + width + 0
        if diagonals[i] >= 0 and i < length1 and diagonals[i + 1] - 1 == diagonals[i]:
            m_count += 1
        elif diagonals[i] == -2:
            m_count = 0
        y = max(y1, y2)
        y = min(y, length2)
        if diagonals[i] >= 0 and i < length1 and diagonals[i + 1] - 1 == diagonals[i] and m_count >= width:
            prev_y = y
        y = max(y, prev_y)
        y = min(y, length2)
        lower_bound.insert(0, y)                                     # :446
    return upper_bound, lower_bound


class TunnelBlock:                                                   # find_anchors.h:51-70
    def __init__(self):
        self.start = [0, 0]
        self.end = [0, 0]

    def size(self):
        return (self.end[0] - self.start[0]) * (self.end[1] - self.start[1])

    def copy(self):
        b = TunnelBlock()
        b.start, b.end = list(self.start), list(self.end)
        return b


def define_tunnel_with_overlapping_hits(hits, sequence1, sequence2, width=15):
    """find_anchors.cpp:633-861 (every hit is a plus-strand hit here).  Returns (upper, lower, empty_blocks) with the blocks
    as (start.x, start.y, end.x, end.y), ascending by size (equal sizes keep their order: the oracle's documented choice)."""
    l1, l2 = len(sequence1), len(sequence2)
    i1 = [i + 1 for i in range(l1) if sequence1[i] != '-']            # :650-662
    i2 = [i + 1 for i in range(l2) if sequence2[i] != '-']
    min_height, max_height = 0, l2                                   # :671-672
    lowest_points = [max_height + 1] * (l1 + 1)                      # :674-677
    highest_points = [min_height - 1] * (l1 + 1)
    for (s1, s2, length, _score) in hits:                            # :680
        for a in range(length):
            if i2[s2 + a] < lowest_points[i1[s1 + a]]:
                lowest_points[i1[s1 + a]] = max(i2[s2 + a], min_height)
            if i2[s2 + a] > highest_points[i1[s1 + a]]:
                highest_points[i1[s1 + a]] = min(i2[s2 + a], max_height)
    # must not go zigzag                                                                                         :696-714
    previous_highest = highest_points[0]
    for i in range(0, l1 + 1):
        if highest_points[i] > min_height:
            if highest_points[i] < previous_highest:
                highest_points[i] = previous_highest
            previous_highest = highest_points[i]
    previous_lowest = lowest_points[l1]
    for i in range(l1, -1, -1):
        if lowest_points[i] < max_height:
            if lowest_points[i] > previous_lowest:
                lowest_points[i] = previous_lowest
            previous_lowest = lowest_points[i]
    # empty blocks                                                                                               :716-746
    empty_blocks = []
    current_block = TunnelBlock()
    for i in range(1, l1 + 1):
        if highest_points[i - 1] >= min_height and highest_points[i] < min_height:
            current_block.start = [i, highest_points[i - 1]]
        elif highest_points[i] >= min_height and highest_points[i - 1] < min_height:
            if lowest_points[i] > current_block.start[1]:
                current_block.end = [i, lowest_points[i]]
                if current_block.size() > 10:
                    empty_blocks.append(current_block.copy())
        elif i == l1 and highest_points[i] < min_height:
            if max_height > current_block.start[1]:
                current_block.end = [i, max_height]
                if current_block.size() > 10:
                    empty_blocks.append(current_block.copy())
    empty_blocks.sort(key=lambda b: b.size())                        # :748 (std::sort; Python's sort is stable)
    # bounds inside the gaps                                                                                     :751-768
    previous_lowest, previous_highest = min_height, max_height
    for i in range(0, l1 + 1):
        if lowest_points[i] >= max_height:
            lowest_points[i] = previous_lowest
        previous_lowest = lowest_points[i]
    for i in range(l1, -1, -1):
        if highest_points[i] <= min_height:
            highest_points[i] = previous_highest
        previous_highest = highest_points[i]
    lowest_points[0] = min_height                                    # :771-772
    highest_points[l1] = max_height
    for i in range(0, l1 + 1):                                       # thicker on the y axis                      :777-786
        if highest_points[i] >= min_height:
            highest_points[i] = min(max_height, highest_points[i] + width)
    for i in range(0, l1 + 1):
        if lowest_points[i] <= max_height:
            lowest_points[i] = max(min_height, lowest_points[i] - width)
    overflow_highest = []                                            # thickness on the x axis                    :790-809
    for i in range(1, l1 + 1):
        if (i + 1 > l1 or highest_points[i] == highest_points[i + 1]) and highest_points[i - 1] < highest_points[i] - 1:
            overflow_highest.append((i, True))
        elif highest_points[i - 1] < highest_points[i] - 1:
            overflow_highest.append((i, False))
    for (i, gapped) in overflow_highest:
        x = i - 1
        while x >= i - width and x >= 0 and highest_points[x] >= min_height:
            if gapped:
                highest_points[x] = max(highest_points[x], highest_points[i])
            else:
                highest_points[x] = max(highest_points[x], highest_points[x + 1] - 1)
            x -= 1
    overflow_lowest = []                                             # :811-830
    for i in range(l1 - 1, -1, -1):
        if (i - 1 < 0 or lowest_points[i] == lowest_points[i - 1]) and lowest_points[i + 1] > lowest_points[i] + 1:
            overflow_lowest.append((i, True))
        elif lowest_points[i + 1] > lowest_points[i] + 1:
            overflow_lowest.append((i, False))
    for (i, gapped) in overflow_lowest:
        x = i + 1
        while x <= i + width and x <= l1 and lowest_points[x] <= max_height:
            if gapped:
                lowest_points[x] = min(lowest_points[x], lowest_points[i])
            else:
                lowest_points[x] = min(lowest_points[x], lowest_points[x - 1] + 1)
            x += 1
    upper = list(lowest_points)                                      # sequence 1 on the y axis                   :843-846
    lower = list(highest_points)
    return upper, lower, [(b.start[0], b.start[1], b.end[0], b.end[1]) for b in empty_blocks]
