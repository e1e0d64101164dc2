"""TEST INFRASTRUCTURE ONLY (oracle) -- an INDEPENDENT definition of the collision predicate's narrow phase.

oracle/collision_ref.c and csrc/collision.hip both evaluate "posed triangle intersects occupied leaf box" with the 13-axis
separating-axis test (Akenine-Moller) in float32.  They are bit-equal to each other, but they share their formulation.  This
module decides the same question a different way, with no separating axes at all:

    the closed triangle and the closed axis-aligned cube intersect  <=>  clipping the triangle (as a convex polygon) against
    the cube's six closed half-spaces (Sutherland-Hodgman) leaves at least one point

in exact rational arithmetic (`fractions.Fraction`; every float is a rational, so float32 inputs are represented exactly) or in
float64 for bulk runs.  tests/test_collision_oracle_cpu.py requires the float32 SAT to agree with it in BOTH directions: exactly
on inputs whose SAT arithmetic is exact (dyadic coordinates -- including every touching configuration: vertex on a face, edge
through a cube edge, coplanar with a face), and outside a stated epsilon band on random float32 inputs.

What this pins: the predicate is the closed-set intersection of FCL's box-vs-triangle narrow phase semantics (contact counts as
collision).  What it cannot pin: FCL's own GJK tolerance at grazing contact (FCL/octomap are absent: PARITY UNPINNED, see
collision_ref.c)."""
from fractions import Fraction


def _clip(poly, axis, sign, bound):
    """Keep the part of convex polygon `poly` (list of 3-tuples) with sign*(p[axis]) <= bound  (closed half-space)."""
    out = []
    n = len(poly)
    for i in range(n):
        p, q = poly[i], poly[(i + 1) % n]
        sp, sq = bound - sign * p[axis], bound - sign * q[axis]          # >= 0 inside
        if sp >= 0:
            out.append(p)
        if (sp >= 0) != (sq >= 0):
            t = sp / (sp - sq)
            out.append(tuple(p[k] + t * (q[k] - p[k]) for k in range(3)))
    return out


def tri_box_intersect(c, h, a, b, d, exact=True):
    """Closed triangle (a,b,d) vs closed cube centre c half-edge h.  exact=True: Fractions; False: float64."""
    conv = (lambda x: Fraction(float(x))) if exact else float
    c = [conv(x) for x in c]; h = conv(h)
    poly = [tuple(conv(x) - c[k] for k, x in enumerate(v)) for v in (a, b, d)]      # cube-centred coordinates
    for axis in range(3):
        for sign in (1, -1):
            poly = _clip(poly, axis, sign, h)
            if not poly:
                return False
    return True
