"""Calendar clock of the simulation - host-side scalar logic that feeds the hot path.

Same public surface and behaviour as the reference's ``Timer`` (grad_june/timer.py:29-157):
the kernels take ``now`` (days since the initial day), ``duration`` (length of the current
shift in days), ``day_type`` and the activity list of the current shift, sorted by the fixed
activity hierarchy (which is also the order in which per-network terms are accumulated).
"""
from __future__ import annotations

import calendar
import datetime
from typing import Mapping, Sequence, Union

import yaml

from .utils import read_date

SECONDS_PER_DAY = 86400.0

#: accumulation order of the infection networks (reference grad_june/timer.py:14-26)
activity_hierarchy = [
    "school", "university", "company", "care_home",
    "pub", "gym", "grocery", "visit", "care_visit", "cinema",
    "household",
]
_RANK = {name: i for i, name in enumerate(activity_hierarchy)}

Shifts = Union[Sequence, Mapping]


class Timer:
    def __init__(
        self,
        initial_day: str = "2020-03-01",
        total_days: int = 10,
        weekday_step_duration: Shifts = (12, 12),
        weekend_step_duration: Shifts = (24,),
        weekday_activities: Shifts = (("school", "household"), ("pub", "household")),
        weekend_activities: Shifts = (("household",),),
    ):
        self.initial_date = read_date(initial_day)
        self.total_days = total_days
        self.weekday_step_duration = weekday_step_duration
        self.weekend_step_duration = weekend_step_duration
        self.weekday_activities = weekday_activities
        self.weekend_activities = weekend_activities
        self.final_date = self.initial_date + datetime.timedelta(days=total_days)
        self.n_timesteps = 0
        self.reset()

    # construction from the YAML schema ------------------------------------------------------
    @classmethod
    def from_file(cls, fpath=None):
        if fpath is None:
            from .defaults import default_parameters

            return cls.from_parameters(default_parameters())
        with open(fpath) as f:
            return cls.from_parameters(yaml.safe_load(f))

    @classmethod
    def from_parameters(cls, params):
        cfg = params["timer"]
        return cls(
            initial_day=cfg["initial_day"],
            total_days=cfg["total_days"],
            weekday_step_duration=cfg["step_duration"]["weekday"],
            weekend_step_duration=cfg["step_duration"]["weekend"],
            weekday_activities=cfg["step_activities"]["weekday"],
            weekend_activities=cfg["step_activities"]["weekend"],
        )

    # state ------------------------------------------------------------------------------------
    def reset(self):
        self.date = self.initial_date
        self.previous_date = self.initial_date
        self.shift = 0
        self.delta_time = datetime.timedelta(hours=self.shift_duration)

    def __next__(self):
        self.previous_date = self.date
        self.date = self.date + self.delta_time
        # a new calendar day restarts the shift counter
        self.shift = 0 if self.date.day != self.previous_date.day else self.shift + 1
        self.delta_time = datetime.timedelta(hours=self.shift_duration)
        self.n_timesteps += 1
        return self.date

    # calendar views ---------------------------------------------------------------------------
    @property
    def is_weekend(self) -> bool:
        return self.date.weekday() >= 5

    @property
    def day_type(self) -> str:
        return "weekend" if self.is_weekend else "weekday"

    @property
    def now(self) -> float:
        return (self.date - self.initial_date).total_seconds() / SECONDS_PER_DAY

    @property
    def duration(self) -> float:
        return self.delta_time.total_seconds() / SECONDS_PER_DAY

    @property
    def day(self) -> int:
        return int(self.now)

    @property
    def day_of_week(self) -> str:
        return calendar.day_name[self.date.weekday()]

    @property
    def date_str(self) -> str:
        return self.date.strftime("%Y-%m-%d")

    @property
    def activities(self):
        return getattr(self, self.day_type + "_activities")[self.shift]

    @property
    def shift_duration(self):
        return getattr(self, self.day_type + "_step_duration")[self.shift]

    def get_activity_order(self):
        """Activities of the current shift, by hierarchy rank (stable for equal ranks)."""
        return sorted(self.activities, key=activity_hierarchy.index)
