"""`Config` -- the user-facing parameter set, field-for-field compatible with the
reference dataclass (/root/reference/ssrs/config.py:9-67: 34 fields + the
`turbine_mrkr_styles` class attribute), so existing scripts that build a
`Config(...)` or call `dataclasses.replace(cfg, ...)` keep working.

The annotations deliberately repeat the reference's (including its oddities,
e.g. `resolution: int = 100.`), because dataclass field order/defaults ARE the
API.  Fields added by this build come last and default to reference behaviour.
"""
import os
from dataclasses import dataclass, fields
from typing import Optional, Tuple

_SECTIONS = (
    ('General settings', ('run_name', 'out_dir', 'max_cores', 'sim_seed', 'sim_mode',
                          'print_verbose')),
    ('Terrain settings', ('southwest_lonlat', 'projected_crs', 'region_width_km',
                          'resolution')),
    ('Uniform mode', ('uniform_winddirn', 'uniform_windspeed')),
    ('Snapshot mode', ('snapshot_datetime',)),
    ('Seasonal mode', ('seasonal_start', 'seasonal_end', 'seasonal_timeofday',
                       'seasonal_count')),
    ('WindToolKit settings', ('wtk_source', 'wtk_orographic_height', 'wtk_thermal_height',
                              'wtk_interp_type')),
    ('Updraft computation', ('thermals_realization_count', 'updraft_threshold',
                             'movement_model')),
    ('Simulating tracks', ('track_direction', 'track_count', 'track_start_region',
                           'track_start_type', 'track_stochastic_nu',
                           'track_dirn_restrict')),
    ('Plotting and wind turbines', ('turbine_minimum_hubheight', 'turbine_mrkr_size',
                                    'fig_height', 'fig_dpi')),
    ('MI355X build', ('save_tracks', 'stepper_path', 'steps_per_launch', 'max_tracks_file_gb', 'hist_safe_tracks')),
)


@dataclass
class Config:
    """Configuration parameters for SSRS simulation """

    # -- general
    run_name: str = 'default'
    out_dir: str = os.path.join(os.path.abspath(os.path.curdir), 'output')
    max_cores: int = 8          # kept for compatibility; tracks run on the GPU
    sim_seed: int = -1          # < 0: unseeded (a fresh seed is drawn per run)
    sim_mode: str = 'uniform'   # uniform | snapshot | seasonal
    print_verbose: bool = False

    # -- terrain
    southwest_lonlat: Tuple[float, float] = (-106.21, 42.78)
    projected_crs: str = 'ESRI:102008'
    region_width_km: Tuple[float, float] = (60., 50.)
    resolution: int = 100.

    # -- uniform mode
    uniform_winddirn: float = 270.   # degrees clockwise from north (270 = westerly)
    uniform_windspeed: float = 10.   # m/s

    # -- snapshot mode
    snapshot_datetime: Tuple[int, int, int, int] = (2010, 6, 17, 13)

    # -- seasonal mode
    seasonal_start: Tuple[int, int] = (3, 20)
    seasonal_end: Tuple[int, int] = (5, 15)
    seasonal_timeofday: str = 'daytime'
    seasonal_count: int = 8

    # -- WIND Toolkit
    wtk_source: str = 'AWS'
    wtk_orographic_height: int = 100
    wtk_thermal_height: int = 100
    wtk_interp_type: str = 'linear'

    # -- updrafts
    thermals_realization_count: bool = 0
    updraft_threshold: float = 0.75
    movement_model: str = 'fluidflow'   # fluidflow | drw

    # -- tracks
    track_direction: float = 0
    track_count: str = 1000
    track_start_region: Tuple[float, float, float, float] = (5, 55, 1, 2)
    track_start_type: str = 'random'    # structured | random
    track_stochastic_nu: float = 1.
    track_dirn_restrict: int = 1

    # -- turbines / plotting (carried for compatibility)
    turbine_minimum_hubheight: float = 50.
    turbine_mrkr_styles = ('1k', '2k', '3k', '4k', '+k', 'xk', '*k', '.k', 'ok')
    turbine_mrkr_size: float = 3.
    fig_height: float = 6.
    fig_dpi: int = 200

    # -- added by the MI355X build (defaults keep the reference's behaviour)
    save_tracks: bool = True            # write <id>_tracks.pkl like the reference
    stepper_path: str = 'auto'          # auto | table | direct
    steps_per_launch: int = 0           # 0 = library default
    max_tracks_file_gb: float = 64.     # refuse a <id>_tracks.pkl larger than this (tracks that wander to
    #                                     max_moves: 1 TB per 100k tracks on a solved 10 m field)
    hist_safe_tracks: int = 250_000     # tracks per sub-batch of a case: (i) histograms of several sub-batches are added up in 64
    #                                     bits; a sub-batch of more than 100 000 tracks is counted in 64 bits inside the library (a trap
    #                                     cell of the solved 10 m field takes 1.7e4 visits per track: 2^32 from ~245 000 tracks on;
    #                                     ssrs_tracks_simulate_h64), and a smaller one whose uint32 counts wrap all the same -- or one
    #                                     with trajectories, which stay on the uint32 raster -- is stepped again as two halves;
    #                                     (ii) ~42 % of such a batch ends up roaming, and the roaming stepper holds ONE block per
    #                                     CU: 250 000 tracks = ~105 000 roaming at first, one round of 512-lane blocks, later
    #                                     ~78 000 in 256-lane blocks: 2.0 s per pass = 1.27e5 tracks/s against 1.0e5 for 140 000
    #                                     (rounds 3-4) and 1.15e5 for 300 000 (a second round of blocks; profiles/r04_roam_fill.txt)

    def __str__(self):
        known = {f.name for f in fields(self)}
        lines = [self.__doc__, '']
        for title, names in _SECTIONS:
            lines.append(f':::: {title}')
            lines.extend(f'{n} = {getattr(self, n)}' for n in names if n in known)
            lines.append('')
        return '\n'.join(lines)
