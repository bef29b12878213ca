"""Default configuration, in the reference's YAML schema (grad_june/configs/default.yaml).

The numbers are the reference's defaults (JUNE leisure attendance, Covasim-style symptom
trajectories); they are stored here as compact per-bin tables and expanded by
:func:`default_parameters` into exactly the nested dict ``yaml.safe_load`` gives for the
reference file, so every ``from_parameters`` accepts either.
"""
from __future__ import annotations

import copy

_LEISURE_BINS = ('0-9', '9-15', '15-19', '19-31', '31-51', '51-66', '66-86', '86-100')

# network -> (weekday male, weekday female, weekend male, weekend female), one value per bin
_LEISURE = {
    'pub': (
        (0.0064, 0.0212, 0.0252, 0.1476, 0.0842, 0.1066, 0.03, 0.0066),
        (0.027000000000000003, 0.0294, 0.0716, 0.10880000000000001, 0.08, 0.0818, 0.020200000000000003, 0.004),
        (0.019, 0.0505, 0.053, 0.1605, 0.131, 0.152, 0.088, 0.0315),
        (0.0215, 0.0405, 0.0705, 0.1255, 0.1155, 0.09, 0.073, 0.03),
    ),
    'cinema': (
        (0.0, 0.0054, 0.0012000000000000001, 0.0032, 0.0018, 0.0018, 0.003, 0.0),
        (0.0224, 0.0066, 0.0198, 0.0066, 0.0028, 0.0062, 0.0016, 0.0),
        (0.0095, 0.0065, 0.0, 0.0, 0.0045, 0.002, 0.005, 0.0),
        (0.0165, 0.007, 0.007, 0.006, 0.0055, 0.005, 0.004, 0.0),
    ),
    'gym': (
        (0.0248, 0.046200000000000005, 0.11739999999999999, 0.0804, 0.0548, 0.0536, 0.0206, 0.0038),
        (0.08080000000000001, 0.07339999999999999, 0.033600000000000005, 0.0426, 0.036, 0.0368, 0.0072, 0.0021999999999999997),
        (0.0925, 0.095, 0.1185, 0.0725, 0.065, 0.073, 0.0445, 0.0),
        (0.037, 0.066, 0.022, 0.0335, 0.0385, 0.0205, 0.0155, 0.002),
    ),
    'visit': (
        (0.4624, 0.343, 0.324, 0.2764, 0.1258, 0.1638, 0.1598, 0.09459999999999999),
        (0.4662, 0.3196, 0.5236, 0.274, 0.146, 0.219, 0.2226, 0.0646),
        (0.562, 0.5535, 0.505, 0.4375, 0.222, 0.228, 0.183, 0.0745),
        (0.7425, 0.657, 0.6075, 0.518, 0.259, 0.2585, 0.2235, 0.101),
    ),
    'grocery': (
        (0.028599999999999997, 0.0392, 0.043, 0.0534, 0.0762, 0.079, 0.0392, 0.013000000000000001),
        (0.0472, 0.0526, 0.0648, 0.0984, 0.124, 0.16299999999999998, 0.0492, 0.0182),
        (0.0535, 0.057, 0.0555, 0.087, 0.1005, 0.1225, 0.0875, 0.041),
        (0.0555, 0.0995, 0.095, 0.1145, 0.133, 0.1405, 0.1145, 0.075),
    ),
}

# overlapping bins on purpose (reference default): key order matters
_CARE_VISIT = {'0-75': 0.0, '75-85': 0.25, '75-100': 0.5}

_DECADES = ('0-10', '10-20', '20-30', '30-40', '40-50', '50-60', '60-70', '70-80', '80-90', '90-100')
_STAGES = ('recovered', 'susceptible', 'exposed', 'infectious', 'symptomatic', 'severe', 'critical', 'dead')
_PROGRESS = {
    'recovered': 0.0,
    'susceptible': 0.0,
    'exposed': 1.0,
    'infectious': (0.5, 0.55, 0.6, 0.65, 0.7, 0.75, 0.8, 0.85, 0.9, 0.9),
    'symptomatic': (0.0005, 0.00165, 0.0072, 0.0208, 0.0343, 0.0765, 0.1328, 0.20655, 0.2457, 0.2457),
    'severe': (3e-05, 8e-05, 0.00036, 0.00104, 0.00216, 0.00933, 0.03639, 0.08923, 0.1742, 0.1742),
    'critical': (2e-05, 2e-05, 0.0001, 0.00032, 0.00098, 0.00265, 0.00766, 0.02439, 0.08292, 0.1619),
}
# LogNormal (loc, scale) of the dwell time before progressing / before recovering
_T_NEXT = {
    'exposed': (1.4513971389473608, 0.32459284597450133),
    'infectious': (-0.16092839609790693, 0.7158750951139896),
    'symptomatic': (1.667557249282718, 0.6625894652794622),
    'severe': (-0.10536051565782628, 1.0107676525947895),
    'critical': (2.278566445372413, 0.42819924356646805),
}
_T_RECOVER = {
    'exposed': (1.4513971389473608, 0.32459284597450133),
    'infectious': (2.0491292307716185, 0.24622067706923975),
    'symptomatic': (2.0491292307716185, 0.24622067706923975),
    'severe': (2.8387344001307495, 0.3381642741066263),
    'critical': (2.8387344001307495, 0.3381642741066263),
}

_LOG_BETA = {'household': -0.4, 'company': -0.3, 'school': -0.3, 'pub': -1.2, 'gym': -1.2, 'grocery': -1.2, 'visit': -1.2, 'cinema': -1.2, 'university': -0.5, 'care_visit': -0.4, 'care_home': -0.4}

_WEEKDAY = ['company', 'school', 'university', 'pub', 'grocery', 'gym', 'cinema', 'visit', 'care_visit', 'care_home', 'household']
_WEEKEND = ['pub', 'grocery', 'gym', 'cinema', 'visit', 'care_visit', 'care_home', 'household']


def _bins(names, values):
    return {b: v for b, v in zip(names, values)}


def _lognormal(table):
    return {k: {"dist": "LogNormal", "loc": loc, "scale": scale} for k, (loc, scale) in table.items()}


def default_parameters(device: str = "cuda:0") -> dict:
    """The default configuration as the nested dict every ``from_parameters`` takes."""
    leisure = {}
    for net, (wd_m, wd_f, we_m, we_f) in _LEISURE.items():
        leisure[net] = {
            "weekday": {"male": _bins(_LEISURE_BINS, wd_m), "female": _bins(_LEISURE_BINS, wd_f)},
            "weekend": {"male": _bins(_LEISURE_BINS, we_m), "female": _bins(_LEISURE_BINS, we_f)},
        }
    leisure["care_visit"] = {d: {"female": dict(_CARE_VISIT), "male": dict(_CARE_VISIT)} for d in ("weekday", "weekend")}
    progress = {}
    for stage, v in _PROGRESS.items():
        progress[stage] = {"0-100": v} if not isinstance(v, tuple) else _bins(_DECADES, v)
    leisure_sd = {k: 0.5 for k in ("pub", "cinema", "gym", "grocery", "visit")}
    params = {
        "title": "grad_june_amd default configuration (reference schema).",
        "system": {"device": device, "random_seed": "random"},
        "data_path": "@grad_june_amd/worlds/world769.npz",
        "save_path": "./example",
        "age_bins_to_save": [0, 18, 65, 100],
        "timer": {
            "total_days": 15,
            "initial_day": "2022-02-01",
            "step_duration": {"weekday": {0: 24}, "weekend": {0: 24}},
            "step_activities": {"weekday": {0: list(_WEEKDAY)}, "weekend": {0: list(_WEEKEND)}},
        },
        "infection_seed": {"log_fraction_initial_cases": -1},
        "networks": {k: {"log_beta": v} for k, v in _LOG_BETA.items()},
        "policies": {
            "interaction": {
                "social_distancing": {
                    1: {"start_date": "2022-02-15", "end_date": "2022-03-15",
                        "beta_factors": {"school": 0.5, "company": 0.5}},
                    2: {"start_date": "2022-03-15", "end_date": "2022-04-15", "beta_factors": dict(leisure_sd)},
                    3: {"start_date": "2023-04-15", "end_date": "2022-05-15", "beta_factors": dict(leisure_sd)},
                }
            }
        },
        "transmission": {
            "max_infectiousness": {"dist": "LogNormal", "loc": 0.0, "scale": 0.5},
            "shape": {"dist": "Normal", "loc": 1.56, "scale": 0.08},
            "rate": {"dist": "Normal", "loc": 0.53, "scale": 0.03},
            "shift": {"dist": "Normal", "loc": -2.12, "scale": 0.1},
        },
        "symptoms": {
            "stages": list(_STAGES),
            "stage_transition_probabilities": progress,
            "stage_transition_times": _lognormal(_T_NEXT),
            "recovery_times": _lognormal(_T_RECOVER),
        },
        "leisure": leisure,
    }
    return copy.deepcopy(params)


def leisure_defaults() -> dict:
    return default_parameters()["leisure"]
