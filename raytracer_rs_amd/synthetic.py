"""Deterministic synthetic scenes, emitted as TOML in the reference's own scene schema (scenes/loader.rs:21-187), so the
reference binary could load them unchanged.

`large_scene()` is the 10 000-object configuration: a checkered floor plus leaves in a bounded three-level group
hierarchy (cell groups -> sub-groups -> leaves), kinds uniform over sphere / cube / closed cone, uniform scales in
[0.25, 0.75] (so |det| >= 0.0156, far above the 1e-5 invertibility threshold of core/matrices.rs:118), no unbounded
primitive inside any group (an unbounded child makes its group's bounds infinite, geometry/bounds.rs:60-69, and
switches culling off).  `random_scene()` is a small scene generator for parity tests that exercises every leaf kind,
every pattern kind, nested groups and reflective / transparent materials.
"""
from __future__ import annotations

from typing import List

MASK = (1 << 64) - 1


class SplitMix64:
    def __init__(self, seed: int):
        self.s = seed & MASK

    def next(self) -> int:
        self.s = (self.s + 0x9E3779B97F4A7C15) & MASK
        z = self.s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & MASK
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & MASK
        return z ^ (z >> 31)

    def uniform(self, lo: float = 0.0, hi: float = 1.0) -> float:
        return lo + (hi - lo) * ((self.next() >> 11) / float(1 << 53))

    def below(self, n: int) -> int:
        return self.next() % n


def _f(v: float) -> str:
    return repr(round(float(v), 6))


def _vec(v) -> str:
    return "[" + ", ".join(_f(x) for x in v) + "]"


def _material_lines(prefix: str, rng: SplitMix64) -> List[str]:
    """40 % diffuse, 30 % glass, 30 % mirror -- the material families of the reference's chapter_14 scene."""
    u = rng.uniform()
    color = [rng.uniform(0.2, 1.0) for _ in range(3)]
    out = [f"[{prefix}.material]"]
    if u < 0.4:
        out += [f"color = {_vec(color)}", f"diffuse = {_f(rng.uniform(0.6, 0.9))}", f"specular = {_f(rng.uniform(0.2, 0.7))}",
                f"reflectivity = {_f(rng.uniform(0.1, 0.3))}"]
    elif u < 0.7:
        k = rng.uniform(0.85, 0.95)
        out += ["color = [1.0, 1.0, 1.0]", "diffuse = 0.1", "specular = 0.9", f"transparency = {_f(k)}",
                f"reflectivity = {_f(k)}", "refractive_index = 1.5"]
    else:
        out += [f"color = {_vec(color)}", "diffuse = 0.3", "specular = 0.8", f"reflectivity = {_f(rng.uniform(0.8, 0.95))}"]
    return out


def _leaf_lines(prefix: str, rng: SplitMix64, cx: float, cz: float, half: float) -> List[str]:
    kind = ("sphere", "cube", "cone")[rng.below(3)]
    s = rng.uniform(0.25, 0.75)
    x = cx + rng.uniform(-half, half)
    z = cz + rng.uniform(-half, half)
    out = [f"[[{prefix}]]", f'type = "{kind}"']
    if kind == "cone":
        out += ["minimum = -1.0", "maximum = 0.0", "closed = true"]
    y = s  # spheres and cubes are centred; the cone spans local y in [-1, 0]: all three rest on the floor at y = s
    out +=[f"[{prefix}.transform]", f"translate = {_vec((x, y, z))}", f"rotate_y = {_f(rng.uniform(0.0, 360.0))}",
            f"scale = {_vec((s, s, s))}"]
    out += _material_lines(prefix, rng)
    return out


def large_scene(width: int = 7680, height: int = 4320, cells: int = 10, subgroups: int = 2, leaves_per_subgroup: int = 25,
                seed: int = 0x5EED0001, extent: float = 20.0) -> str:
    """cells x cells top-level groups over x,z in [-extent, extent], each with subgroups x subgroups sub-groups of
    `leaves_per_subgroup` leaves: 10 x 10 x 4 x 25 = 10 000 leaves with the defaults."""
    rng = SplitMix64(seed)
    L: List[str] = ["# synthetic scene (generated; see raytracer_rs_amd/synthetic.py)", "[camera]", f"width = {width}",
                    f"height = {height}", "fov = 70.0", "from = [0.0, 6.0, -18.0]", "to = [0.0, 0.5, 0.0]",
                    "up = [0.0, 1.0, 0.0]", "", "[light]", "position = [-10.0, 10.0, -10.0]", "intensity = [1.0, 1.0, 1.0]", "",
                    "[[objects]]", 'type = "plane"', "[objects.material]", "ambient = 0.2", "diffuse = 0.7", "specular = 0.2",
                    "reflectivity = 0.1", "[objects.material.pattern]", 'type = "checkers"', "color_a = [0.15, 0.15, 0.15]",
                    "color_b = [0.85, 0.85, 0.85]", ""]
    cell = 2.0 * extent / cells
    sub = cell / subgroups
    for i in range(cells):
        for j in range(cells):
            cx = -extent + (i + 0.5) * cell
            cz = -extent + (j + 0.5) * cell
            L += ["[[objects]]", 'type = "group"', "[objects.transform]", f"translate = {_vec((cx, 0.0, cz))}"]
            for a in range(subgroups):
                for b in range(subgroups):
                    sx = -cell / 2 + (a + 0.5) * sub
                    sz = -cell / 2 + (b + 0.5) * sub
                    L += ["[[objects.children]]", 'type = "group"', "[objects.children.transform]",
                          f"translate = {_vec((sx, 0.0, sz))}"]
                    for _ in range(leaves_per_subgroup):
                        L += _leaf_lines("objects.children.children", rng, 0.0, 0.0, sub / 2 - 0.75)
            L.append("")
    return "\n".join(L) + "\n"


def random_scene(seed: int, width: int = 64, height: int = 48, n_top: int = 6) -> str:
    """A small scene for parity tests: every leaf kind (bounded and unbounded, open and closed), every pattern kind
    with pattern transforms, rotations about all axes, non-uniform scales, groups nested two deep with inherited
    materials, and reflective / transparent / both materials."""
    rng = SplitMix64(seed)
    L: List[str] = ["[camera]", f"width = {width}", f"height = {height}", f"fov = {_f(rng.uniform(50, 80))}",
                    f"from = {_vec((rng.uniform(-2, 2), rng.uniform(1.5, 4), -9.0))}", "to = [0.0, 0.8, 0.0]",
                    "up = [0.0, 1.0, 0.0]", "[light]", f"position = {_vec((rng.uniform(-10, -4), 10.0, -10.0))}",
                    f"intensity = {_vec((1.0, rng.uniform(0.8, 1.0), rng.uniform(0.8, 1.0)))}"]
    kinds = ["sphere", "cube", "cylinder", "cone", "plane"]
    patterns = ["stripe", "gradient", "ring", "checkers"]

    def material(prefix: str) -> List[str]:
        u = rng.uniform()
        out = [f"[{prefix}.material]", f"color = {_vec([rng.uniform(0.1, 1.0) for _ in range(3)])}",
               f"ambient = {_f(rng.uniform(0.0, 0.3))}", f"diffuse = {_f(rng.uniform(0.2, 0.9))}",
               f"specular = {_f(rng.uniform(0.0, 0.9))}", f"shininess = {_f(rng.uniform(10, 300))}"]
        if u < 0.25:
            out += [f"reflectivity = {_f(rng.uniform(0.2, 0.9))}"]
        elif u < 0.45:
            out += [f"transparency = {_f(rng.uniform(0.3, 1.0))}", f"refractive_index = {_f(rng.uniform(1.0, 2.0))}"]
        elif u < 0.65:
            out += [f"reflectivity = {_f(rng.uniform(0.2, 0.9))}", f"transparency = {_f(rng.uniform(0.3, 1.0))}",
                    f"refractive_index = {_f(rng.uniform(0.7, 2.0))}"]
        if rng.uniform() < 0.5:
            out += [f"[{prefix}.material.pattern]", f'type = "{patterns[rng.below(4)]}"',
                    f"color_a = {_vec([rng.uniform(0, 1) for _ in range(3)])}",
                    f"color_b = {_vec([rng.uniform(0, 1) for _ in range(3)])}",
                    f"scale = {_vec([rng.uniform(0.2, 1.0) for _ in range(3)])}", f"rotate_z = {_f(rng.uniform(0, 90))}"]
        return out

    def transform(prefix: str, spread: float) -> List[str]:
        out = [f"[{prefix}.transform]",
               f"translate = {_vec((rng.uniform(-spread, spread), rng.uniform(0.2, 1.5), rng.uniform(-spread, spread)))}"]
        if rng.uniform() < 0.6:
            out.append(f"rotate_x = {_f(rng.uniform(-60, 60))}")
        if rng.uniform() < 0.6:
            out.append(f"rotate_y = {_f(rng.uniform(0, 360))}")
        if rng.uniform() < 0.4:
            out.append(f"rotate_z = {_f(rng.uniform(-45, 45))}")
        out.append(f"scale = {_vec([rng.uniform(0.3, 1.1) for _ in range(3)])}")
        return out

    def leaf(prefix: str, spread: float, allow_plane: bool) -> List[str]:
        kind = kinds[rng.below(5 if allow_plane else 4)]
        out = [f"[[{prefix}]]", f'type = "{kind}"']
        if kind in ("cylinder", "cone") and rng.uniform() < 0.8:
            lo = rng.uniform(-1.0, 0.0)
            out += [f"minimum = {_f(lo)}", f"maximum = {_f(lo + rng.uniform(0.5, 1.5))}"]
            if rng.uniform() < 0.7:
                out.append("closed = true")
        if kind != "plane":
            out += transform(prefix, spread)
        if rng.uniform() < 0.8:
            out += material(prefix)
        return out

    L += ["[[objects]]", 'type = "plane"', "[objects.material]", "reflectivity = 0.3", "[objects.material.pattern]",
          'type = "checkers"', "color_a = [0.2, 0.2, 0.2]", "color_b = [0.9, 0.9, 0.9]"]
    for _ in range(n_top):
        if rng.uniform() < 0.5:
            L += leaf("objects", 3.5, False)
        else:
            L += ["[[objects]]", 'type = "group"', "[objects.transform]",
                  f"translate = {_vec((rng.uniform(-3, 3), 0.0, rng.uniform(-2, 4)))}",
                  f"rotate_y = {_f(rng.uniform(0, 360))}", f"scale = {_vec([rng.uniform(0.6, 1.2)] * 3)}"]
            if rng.uniform() < 0.6:
                L += material("objects")
            for _ in range(1 + rng.below(3)):
                if rng.uniform() < 0.35:
                    L += ["[[objects.children]]", 'type = "group"', "[objects.children.transform]",
                          f"translate = {_vec((rng.uniform(-1, 1), rng.uniform(0, 1), rng.uniform(-1, 1)))}",
                          f"rotate_z = {_f(rng.uniform(-30, 30))}"]
                    for _ in range(1 + rng.below(3)):
                        L += leaf("objects.children.children", 1.0, False)
                else:
                    L += leaf("objects.children", 1.5, False)
    return "\n".join(L) + "\n"


def flat_scene(seed: int) -> str:
    """A world WITHOUT groups and with at most twelve objects (the table-of-leaves kernels): mostly glass and mirror
    balls, some stretched or turned, a floor, now and then a cube or a cylinder; the light anywhere -- above the scene,
    close to a ball, inside one."""
    rng = SplitMix64(seed)
    u = rng.uniform
    n = 3 + rng.below(9)
    centres = [(u(-3, 3), u(0.3, 2.0), u(-3, 3)) for _ in range(n)]
    where = rng.below(4)
    if where == 0:
        light = (u(-10, 10), u(4, 10), u(-10, -2))
    elif where == 1:  # inside or just outside some ball
        c = centres[rng.below(n)]
        light = (c[0] + u(-0.8, 0.8), c[1] + u(-0.8, 0.8), c[2] + u(-0.8, 0.8))
    elif where == 2:
        light = (u(-4, 4), u(0.2, 3), u(-4, 4))
    else:
        light = (u(-10, -4), 10.0, -10.0)
    L = ["[camera]", "width = 160", "height = 100", f"fov = {u(50, 80)}", f"from = [{u(-2, 2)}, {u(1.0, 4.0)}, -8.0]",
         "to = [0.0, 0.8, 0.0]", "up = [0.0, 1.0, 0.0]", "[light]", f"position = [{light[0]}, {light[1]}, {light[2]}]",
         "intensity = [1.0, 0.95, 0.9]", "[[objects]]", 'type = "plane"', "[objects.material]", "color = [0.7, 0.7, 0.7]",
         f"reflectivity = {u(0.0, 0.5)}", f"transparency = {0.3 if rng.below(3) == 0 else 0.0}", "refractive_index = 1.4"]
    for k in range(n):
        kind = ["sphere", "sphere", "sphere", "sphere", "cube", "cylinder"][rng.below(6)]
        s = u(0.2, 1.2)
        scale = (s, s, s) if rng.below(3) else (s * u(0.5, 1.5), s * u(0.5, 1.5), s * u(0.5, 1.5))
        L += ["[[objects]]", f'type = "{kind}"']
        if kind == "cylinder":
            L += ["minimum = -1.0", "maximum = 1.0", f"closed = {'true' if rng.below(2) else 'false'}"]
        L += ["[objects.transform]", f"scale = [{scale[0]}, {scale[1]}, {scale[2]}]"]
        if rng.below(3) == 0:
            L += [f"rotate_y = {u(0, 180)}", f"rotate_x = {u(0, 90)}"]
        L += [f"translate = [{centres[k][0]}, {centres[k][1]}, {centres[k][2]}]", "[objects.material]",
              f"color = [{u(0.2, 1)}, {u(0.2, 1)}, {u(0.2, 1)}]", f"ambient = {u(0, 0.2)}", f"diffuse = {u(0.1, 0.8)}",
              f"specular = {u(0, 0.9)}", f"shininess = {float(10 + rng.below(290))}"]
        m = rng.below(4)
        if m >= 1:
            L += [f"transparency = {u(0.4, 1.0)}", f"refractive_index = {u(1.0, 1.8)}"]
        if m >= 2:
            L += [f"reflectivity = {u(0.2, 0.9)}"]
    return "\n".join(L) + "\n"
