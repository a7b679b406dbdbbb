"""Names user code imports from the reference's goal-wrapper module
(carl/envs/brax/brax_walker_goal_wrapper.py:16-50): the compass codes a ``target_direction`` feature can
take and their spoken names.  The wrapper itself -- goal position from (direction, distance), position
integrated from the observed planar velocity, progress reward, success inside the radius (:69-140) -- is
not a Python wrapper here: it is an epilogue of the Brax step kernel (``carl_brax_sys_t::goal_mode``,
DESIGN.md section 5), switched on by ``CARLBraxEnv`` when the goal features vary across the contexts.
The language goal of ``BraxLanguageWrapper`` (:143-181) is a host-side sentence:
``CARLBraxEnv.describe_goal`` / ``use_language_goals=True``."""
from __future__ import annotations

from carl_amd.envs.brax.feature_tables import DIRECTIONS

_COMPASS = {"1": "north", "2": "east", "3": "south", "4": "west"}

# codes read digit by digit: 1 -> north, 12 -> north east, 334 -> south south west
directions = list(DIRECTIONS)
DIRECTION_NAMES = {code: " ".join(_COMPASS[d] for d in str(code)) for code in directions}
