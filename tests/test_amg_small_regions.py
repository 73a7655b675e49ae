"""min_mask_region_area > 0 (VERDICT r4 missing #4): the host clean-up of SAM's automatic mask generator -- `remove_small_regions` /
`postprocess_small_regions` of sam2 / segment_anything `utils.amg`, restated (UNPINNED: the packages are not available offline) -- against a
brute-force flood fill written here, on random masks."""
import numpy as np

from ovo_amd.entities.sam_amg import postprocess_small_regions, remove_small_regions


def _components(working):
    """8-connected components by explicit flood fill: list of pixel lists."""
    h, w = working.shape
    seen = np.zeros_like(working, bool)
    out = []
    for y in range(h):
        for x in range(w):
            if working[y, x] and not seen[y, x]:
                stack, comp = [(y, x)], []
                seen[y, x] = True
                while stack:
                    cy, cx = stack.pop()
                    comp.append((cy, cx))
                    for dy in (-1, 0, 1):
                        for dx in (-1, 0, 1):
                            ny, nx = cy + dy, cx + dx
                            if 0 <= ny < h and 0 <= nx < w and working[ny, nx] and not seen[ny, nx]:
                                seen[ny, nx] = True
                                stack.append((ny, nx))
                out.append(comp)
    return out


def _reference(mask, area, mode):
    holes = mode == "holes"
    working = np.logical_xor(holes, mask)
    comps = _components(working)
    small = [c for c in comps if len(c) < area]
    if not small:
        return mask.copy(), False
    out = mask.copy()
    if holes:                                                      # small background components become foreground
        for c in small:
            for y, x in c:
                out[y, x] = True
        return out, True
    keep = [c for c in comps if len(c) >= area]
    if not keep:
        keep = [max(comps, key=len)]                               # (ties: any largest; the fixtures below have none)
    out[:] = False
    for c in keep:
        for y, x in c:
            out[y, x] = True
    return out, True


def test_remove_small_regions_vs_flood_fill():
    rng = np.random.default_rng(0)
    for trial in range(40):
        h, w = rng.integers(8, 40, size=2)
        mask = rng.random((h, w)) < rng.choice([0.2, 0.5, 0.8])
        if trial % 4 == 0:                                         # a big blob with pinholes and specks around it
            mask[:] = False
            mask[2:h - 2, 2:w - 2] = True
            mask[rng.integers(3, h - 3, 5), rng.integers(3, w - 3, 5)] = False
            mask[0, 0] = mask[h - 1, w - 1] = True
        for mode in ("holes", "islands"):
            for area in (1, 2, 4, 9, 30):
                got, ch = remove_small_regions(mask, area, mode)
                ref, rch = _reference(mask, area, mode)
                sizes = sorted(len(c) for c in _components(np.logical_xor(mode == "holes", mask)))
                if mode == "islands" and sizes and sizes[-1] < area and sizes.count(sizes[-1]) > 1:
                    continue                                       # several largest islands, all below the threshold: the survivor is a numbering accident
                assert ch == rch and np.array_equal(got, ref), (trial, mode, area)


def test_remove_small_regions_edge_cases():
    empty = np.zeros((6, 7), bool)
    got, ch = remove_small_regions(empty, 5, "islands")
    assert not ch and not got.any()
    full = np.ones((6, 7), bool)
    got, ch = remove_small_regions(full, 5, "holes")
    assert not ch and got.all()
    speck = np.zeros((6, 7), bool)
    speck[2, 3] = True                                             # the only island is below the threshold: it stays (largest)
    got, ch = remove_small_regions(speck, 5, "islands")
    assert ch and np.array_equal(got, speck)
    diag = np.zeros((5, 5), bool)
    diag[0, 0] = diag[1, 1] = diag[2, 2] = True                    # 8-connectivity: one island of three pixels
    got, ch = remove_small_regions(diag, 3, "islands")
    assert not ch and np.array_equal(got, diag)


def test_postprocess_small_regions_prefers_unchanged_masks():
    H, W = 24, 24
    a = np.zeros((H, W), bool); a[2:12, 2:12] = True               # clean square
    b = a.copy(); b[20, 20] = True                                 # the same square + a speck: cleaned it DUPLICATES a
    c = np.zeros((H, W), bool); c[14:22, 4:20] = True; c[17, 10] = False      # a pinhole
    boxes = np.array([[2, 2, 11, 11], [2, 2, 20, 20], [4, 14, 19, 21]], np.int32)
    masks, nb, keep, changed = postprocess_small_regions(np.stack([b, a, c]), boxes[[1, 0, 2]], 4, 0.7)
    # the cleaned b equals a: the box NMS keeps the unchanged one (score 1) and drops the changed duplicate
    assert sorted(keep.tolist()) == [1, 2]
    assert np.array_equal(masks[list(keep).index(1)], a) and not changed[list(keep).index(1)]
    k2 = list(keep).index(2)
    assert changed[k2] and masks[k2][17, 10] and nb[k2].tolist() == [4, 14, 19, 21]
    none = postprocess_small_regions(np.zeros((0, H, W), bool), np.zeros((0, 4), np.int32), 4, 0.7)
    assert len(none[0]) == 0 and len(none[2]) == 0
