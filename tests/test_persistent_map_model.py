"""The ALGORITHM of the persistent submap (open3d_slam_amd/csrc/map_kernels.hpp) restated in plain Python and held to the
oracle's re-binning on the CPU.

Submap::insertScan is `map += T * scan; voxelizeWithinCroppingVolume(map)` (helpers.cpp:115-183): the reference re-bins the whole map at
every scan and its array becomes [points outside the volume, in their previous order | one mean per voxel inside it].  The device keeps
the map in slots and touches only the voxels a scan falls into; what makes that the same ARRAY in the end is a claim about orders:

  * a slot's place in the array after insertion t is a function of its history alone -- the last insertion whose volume held it, the
    voxel key it was binned under then -- (pm_view_key), and
  * when several old members of a voxel come to lie inside the volume together, summing them in the order of the array before this
    insertion (pm_view_key at t - 1), the scan's points after them, gives the reference's mean bit for bit.

The model below is that claim and nothing else: slots with histories, a list of volumes, no hash, no index, no lists (every live slot is
looked at at every insertion -- the device's lists and chains are how it avoids that, not what it computes).  It is run against
orc_voxelize_within_volume (the oracle, pinned to the reference's compiled sources) over out-and-back paths with a small volume: points
leave, pass through for a while and re-enter, several old members of one voxel merge, scan points beyond the volume join unmerged,
normals are re-normalised every time they pass through the volume (a mean that rounding puts across a voxel face keeps the key it was
binned under until the next insertion re-bins it: the same rule, rare in binary64).  Sizes a pure-Python loop finishes
in seconds; the GPU tests hold the kernels to the same arrays at full size (test_persistent_map_is_bitwise_the_array_form)."""
import math

import numpy as np
import pytest

from open3d_slam_amd import synthetic as syn

RAW = 1 << 63


def pack_key(p, inv):  # cloud_kernels.hpp pack_key(floor(p * inv)): z, y, x from the most significant bits down
    kx, ky, kz = (int(math.floor(p[a] * inv)) for a in range(3))
    m = 0x1FFFFF
    return (((kz + (1 << 20)) & m) << 42) | (((ky + (1 << 20)) & m) << 21) | ((kx + (1 << 20)) & m)


def contains(crop, p):  # croppers.cpp:121-124 MinMaxRadius, as the oracle's within_volume
    c, rmin, rmax = crop
    d = math.sqrt((p[0] - c[0]) ** 2 + (p[1] - c[1]) ** 2 + (p[2] - c[2]) ** 2)
    return rmin <= d <= rmax


def mean_of(points, normals):  # AccumulatedPoint (helpers.cpp:30-73) over the members in the given order; .normalized() of the mean normal
    s, q, cnt = [0.0, 0.0, 0.0], [0.0, 0.0, 0.0], 0
    for p, n in zip(points, normals):
        for a in range(3):
            s[a] += p[a]
        if not (math.isnan(n[0]) or math.isnan(n[1]) or math.isnan(n[2])):
            for a in range(3):
                q[a] += n[a]
        cnt += 1
    p = [s[a] / cnt for a in range(3)]
    n = [q[a] / cnt for a in range(3)]
    z = (n[0] * n[0] + n[1] * n[1]) + n[2] * n[2]
    if z > 0.0:
        r = math.sqrt(z)
        n = [n[a] / r for a in range(3)]
    return p, n


class Slot:
    __slots__ = ("p", "n", "st", "ok", "dead")

    def __init__(self, p, n, st, ok):
        self.p, self.n, self.st, self.ok, self.dead = list(p), list(n), st, ok, False


class PersistentMapModel:
    """map_kernels.hpp without its data structures: what an insertion does to the slots, and where a slot stands in the array"""

    def __init__(self, pts, nrm, n_pass, voxel):
        self.inv = 1.0 / voxel
        self.slots = [Slot(p, n, 0, pack_key(p, self.inv)) for p, n in zip(pts, nrm)]  # pm_enter_kernel
        self.np_base, self.n_base = n_pass, len(pts)
        self.hist = [None]  # hist[t]: the volume of insertion t
        self.merges_of_several = self.walks = 0  # (what a run exercised)

    def in_block(self, s, t):  # pm_in_block: "the slot belonged to the voxel block of the array after insertion t"
        sl = self.slots[s]
        if t == 0:
            return self.np_base <= s < self.n_base
        if sl.st == t:
            return not (sl.ok & RAW)
        return contains(self.hist[t], sl.p)

    def view_key(self, s, t_ref):  # pm_view_key
        sl = self.slots[s]
        kp = pack_key(sl.p, self.inv)
        if self.in_block(s, t_ref):
            return (1 << 63, s if t_ref == 0 else (sl.ok if sl.st == t_ref else kp))
        tau = 0
        for t in range(t_ref - 1, max(sl.st, 0) - 1, -1):
            if self.in_block(s, t):
                tau = t + 1  # it belonged to the voxel block of insertion t and left with insertion t + 1
                break
        if tau > 0:
            t = tau - 1
            return (tau << 1, s if t == 0 else (sl.ok if sl.st == t else kp))
        if (sl.ok & RAW) and sl.st > 0:  # inserted outside the volume at insertion st and never inside since
            return ((sl.st << 1) | 1, sl.ok & ~RAW)
        return (0, s)  # a pass-through point of the base that has not been inside since

    def insert(self, scan_p, scan_n, crop):
        t = len(self.hist)
        groups, outside = {}, []
        for i, p in enumerate(scan_p):  # pm_place_kernel / vox_order_kernel: the scan's points by voxel, in scan order
            if contains(crop, p):
                groups.setdefault(pack_key(p, self.inv), []).append(i)
            else:
                outside.append(i)
        old = {}
        for s, sl in enumerate(self.slots):  # (the device finds them through the voxel hash and the multi list)
            if not sl.dead and contains(crop, sl.p):
                old.setdefault(pack_key(sl.p, self.inv), []).append(s)
        for key in sorted(set(groups) | set(old)):
            olds, idx = old.get(key, []), groups.get(key, [])
            if not idx and len(olds) == 1:  # alone and untouched: its point is the mean of one (exact), its normal is re-normalised (pm_misc_kernel)
                sl = self.slots[olds[0]]
                _, sl.n = mean_of([sl.p], [sl.n])
                continue
            if len(olds) > 1:  # pm_merge_kernel: in the order of the array before this insertion
                self.merges_of_several += 1
                self.walks += sum(1 for s in olds if not self.in_block(s, t - 1))
                olds = sorted(olds, key=lambda s: (self.view_key(s, t - 1), s))
            p, n = mean_of([self.slots[s].p for s in olds] + [scan_p[i] for i in idx], [self.slots[s].n for s in olds] + [scan_n[i] for i in idx])
            if olds:
                target = self.slots[olds[0]]
                for s in olds[1:]:
                    self.slots[s].dead = True
            else:
                target = Slot(p, n, t, key)
                self.slots.append(target)
            target.p, target.n, target.st, target.ok = p, n, t, key  # (pm_store: binned under `key`, wherever rounding put the mean)
        for i in outside:  # scan points beyond the volume join the map as they were placed (pm_misc_kernel)
            self.slots.append(Slot(scan_p[i], scan_n[i], t, RAW | i))
        self.hist.append(crop)

    def array(self):  # pm_exit_t: the live slots by their view keys (ties by slot number: the sort is stable)
        t_last = len(self.hist) - 1
        keyed = sorted(((self.view_key(s, t_last), s) for s, sl in enumerate(self.slots) if not sl.dead))
        self.order = [s for _, s in keyed]  # slot of every array position (carve)
        pts = np.array([self.slots[s].p for _, s in keyed]).reshape(-1, 3)
        nrm = np.array([self.slots[s].n for _, s in keyed]).reshape(-1, 3)
        return pts, nrm, sum(1 for k, _ in keyed if k[0] < (1 << 63))


    def carve(self, removed):  # pm_carve_*: the points that go are marked dead where they are; the survivors' order is their histories'
        for pos in np.flatnonzero(removed):
            self.slots[self.order[pos]].dead = True


def reference_step(oracle, pts, nrm, scan_p, scan_n, crop_abi, crop, voxel):
    """map += scan; voxelizeWithinCroppingVolume -- the oracle's means (first-occurrence order), the voxel block put in the order of the
    voxel keys the points were binned under (the array form of the backend: include/o3ds_backend.h o3ds_map_insert_scan)"""
    cat_p, cat_n = np.vstack([pts, scan_p]), np.vstack([nrm, scan_n])
    out_p, out_n, n_pass = oracle.voxelize_within_volume(cat_p, cat_n, voxel, crop_abi)
    inv = 1.0 / voxel
    keys, seen = [], set()
    for p in cat_p:
        if contains(crop, p):
            k = pack_key(p, inv)
            if k not in seen:
                seen.add(k)
                keys.append(k)
    assert len(keys) == len(out_p) - n_pass
    order = np.argsort(np.array(keys, dtype=np.uint64), kind="stable")
    out_p = np.vstack([out_p[:n_pass], out_p[n_pass:][order]])
    out_n = np.vstack([out_n[:n_pass], out_n[n_pass:][order]])
    return out_p, out_n, n_pass


def _scans(oracle, n_frames, n_az, ghosts=()):
    """ghosts: frames whose scan also sees a small obstacle 3 m ahead that later scans look straight through (what space carving removes)"""
    scene = syn.make_scene()
    out = []
    for k in range(n_frames):
        t = k if k < n_frames // 2 else n_frames - 1 - k  # out and back: the volume returns over what it left behind
        T = syn.make_pose([1.5 * t, 0.4 * t, 0.0], [0.0, 0.0, 4.0 * t])
        raw = syn.vlp16_scan(scene, T, frame=k, n_az=n_az)
        v = oracle.voxel_down_sample(raw, 0.25)
        n = oracle.estimate_normals(v, 2.0, 10)
        if k in ghosts:
            g = np.array([[3.0 + 0.05 * a, -0.3 + 0.15 * b, 0.2 + 0.15 * c] for a in range(2) for b in range(5) for c in range(4)])
            v = np.vstack([v, g])
            n = np.vstack([n, np.tile([-1.0, 0.0, 0.0], (len(g), 1))])
        out.append((oracle.transform_points(v, T), oracle.transform_normals(n, T), T))
    return out


@pytest.mark.parametrize("rebase_at", [(), (5,), (2, 9)])
def test_the_persistent_form_is_the_reference_s_array(oracle, rebase_at):
    """16 insertions out and back, map voxel 0.4 m, volume of 7 m around the sensor (most of a scan lies beyond it).  The model never
    re-bins; looked at after every insertion it must be the oracle's array, byte for byte -- continued from its slots, or (rebase_at)
    re-entered from the array as the device does after a fold (a new base: its pass-through block and its voxel block are told apart by
    the slot number alone)."""
    from oracle import pyoracle

    voxel, rmax = 0.4, 7.0
    scans = _scans(oracle, 16, 48)
    ref_p, ref_n = np.zeros((0, 3)), np.zeros((0, 3))
    model = None
    stats = {"merges_of_several": 0, "members_ordered_by_their_history": 0, "outside": 0, "dead": 0, "relinked": 0}
    for k, (sp, sn, T) in enumerate(scans):
        centre = [float(x) for x in T[:3, 3]]
        crop = (centre, 0.0, rmax)
        crop_abi = pyoracle.make_crop(pyoracle.CROP_MIN_MAX_RADIUS, center=centre, rmin=0.0, rmax=rmax)
        ref_p, ref_n, ref_np = reference_step(oracle, ref_p, ref_n, sp, sn, crop_abi, crop, voxel)
        if model is None or k in rebase_at:  # enter (or re-enter) the persistent form from the array
            if model is None:
                model = PersistentMapModel(ref_p, ref_n, ref_np, voxel)
                continue  # (the first insertion of a map is the array form's on the device too)
            got = model.array()
            stats["merges_of_several"] += model.merges_of_several
            stats["members_ordered_by_their_history"] += model.walks
            model = PersistentMapModel(got[0], got[1], got[2], voxel)
        before = len(model.slots)
        model.insert([list(map(float, p)) for p in sp], [list(map(float, n)) for n in sn], crop)
        stats["outside"] += sum(1 for sl in model.slots[before:] if sl.ok & RAW)
        got_p, got_n, got_np = model.array()
        assert got_np == ref_np and len(got_p) == len(ref_p), (k, got_np, ref_np, len(got_p), len(ref_p))
        assert got_p.tobytes() == ref_p.tobytes(), k
        assert got_n.tobytes() == ref_n.tobytes(), k
    stats["merges_of_several"] += model.merges_of_several
    stats["members_ordered_by_their_history"] += model.walks
    stats["dead"] = sum(1 for sl in model.slots if sl.dead)
    stats["relinked"] = sum(1 for sl in model.slots if not sl.dead and not (sl.ok & RAW) and sl.st > 0 and pack_key(sl.p, model.inv) != sl.ok)
    # the run exercised what it is meant to: members that died in merges, scan points that joined outside the volume
    print(stats)
    assert stats["outside"] > 100 and stats["dead"] > 20 and stats["merges_of_several"] > 20, stats
    assert len(ref_p) > 1500


def test_carving_the_persistent_form_in_place_keeps_the_survivors_in_order(oracle):
    """Submap::carve (Submap.cpp:109-125) removes points from the array and keeps the rest in order.  On the persistent form the points
    that go are marked dead in their slots (pm_carve_apply_kernel) and nothing else happens: the survivors' places are functions of their
    histories, not of an array.  Carved before insertions 6, 9 and 13 of the out-and-back run (the rays of the scan being inserted,
    inside the volume of the previous insertion, as Submap::insertScan does it), the model must stay the oracle's array."""
    from oracle import pyoracle

    voxel, rmax = 0.4, 7.0
    scans = _scans(oracle, 16, 48, ghosts=(4, 5, 7, 8, 11, 12))
    ref_p, ref_n = np.zeros((0, 3)), np.zeros((0, 3))
    model, prev_crop, carved = None, None, 0
    for k, (sp, sn, T) in enumerate(scans):
        centre = [float(x) for x in T[:3, 3]]
        crop = (centre, 0.0, rmax)
        crop_abi = pyoracle.make_crop(pyoracle.CROP_MIN_MAX_RADIUS, center=centre, rmin=0.0, rmax=rmax)
        if k in (6, 9, 13):
            subset = np.array([i for i, p in enumerate(ref_p) if contains(prev_crop, p)], dtype=np.int64)
            gone = oracle.carve_flags(sp, centre, ref_p, ref_n, subset, voxel=voxel, max_length=20.0, truncation=0.1, min_dot=0.5)
            assert gone.any() and not gone.all()
            carved += int(gone.sum())
            model.array()
            model.carve(gone)
            ref_p, ref_n = ref_p[~gone], ref_n[~gone]
            got_p, got_n, _ = model.array()
            assert got_p.tobytes() == ref_p.tobytes() and got_n.tobytes() == ref_n.tobytes(), ("after the carve", k)
        ref_p, ref_n, ref_np = reference_step(oracle, ref_p, ref_n, sp, sn, crop_abi, crop, voxel)
        prev_crop = crop
        if model is None:
            model = PersistentMapModel(ref_p, ref_n, ref_np, voxel)
            continue
        model.insert([list(map(float, p)) for p in sp], [list(map(float, n)) for n in sn], crop)
        got_p, got_n, got_np = model.array()
        assert got_np == ref_np and got_p.tobytes() == ref_p.tobytes() and got_n.tobytes() == ref_n.tobytes(), k
    assert carved > 10, carved
