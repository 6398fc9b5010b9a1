"""The float-error Bresenham walk (W12m/bresenham.py:45-55) ends in the cell of its other end.

The byte-window ray cast (k_grid_update_owner8, csrc/grid_kernels.hip) leaves both end cells of a ray
out of its walk: one is the scan's origin (counted once for all rays), the other takes the hit.  That
is the reference's path only if the walk - a float64 running sum of dy/dx, not integer Bresenham -
really arrives at the other end after dx steps.  Checked exhaustively for every (dx, dy) up to the
longest ray that kernel accepts (kOwn8MaxLen = 2048 cells; longer rays go to the general kernel)."""
from oracle import c_oracle as co


def test_float_walk_arrives_at_the_other_end_up_to_2100_cells():
    assert co.walk_end_mismatches(2100) == 0
