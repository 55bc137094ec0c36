"""The model structure entry points of the reference (smartpy/structure.py:30-264), executed on the GPU.

`run`, `run_all_steps` and `run_one_step` keep the reference's names, argument meaning, return values and error
behaviour; their bodies hand the work to the HIP engine through the C ABI (include/smart_amd.h).  `run` also
accepts a whole parameter matrix [N, 10] instead of one 10-vector: that is the batched form the Monte-Carlo layer
uses (one launch for the ensemble instead of one Python call per sample, montecarlo.py:153-154,179-186).
"""
import numpy as np

from . import engine

#: the step function never runs in Python here; kept so that code probing `structure.smart_in_cpp` keeps working
smart_in_cpp = True
smart_on_gpu = True

model_outputs = ['Q_aeva', 'Q_ove', 'Q_dra', 'Q_int', 'Q_sgw', 'Q_dgw', 'Q_out']
model_states = ['V_ove', 'V_dra', 'V_int', 'V_sgw', 'V_dgw',
                'V_ly1', 'V_ly2', 'V_ly3', 'V_ly4', 'V_ly5', 'V_ly6', 'V_river']
model_variables = model_outputs + model_states       # structure.py:78-82


def warm_up_length(warm_up_days, delta_sec, simu_length):
    """structure.py:87-95."""
    length = int(warm_up_days * 86400 / delta_sec)
    if length > simu_length:
        raise Exception("The warm-up duration (i.e. {} days) cannot exceed the length of the simulation period "
                        "because the beginning of the simulation period is used as made-up warm-up data for the "
                        "sake of model states initialisation. Please specify another warm-up duration to comply "
                        "with this requirement, or consider using actual warm-up data at the beginning of the "
                        "simulation period and set the warm-up period to 0.".format(warm_up_days))
    return length


def run(area_m2, delta, nd_rain, nd_peva, nd_parameters, extra, timeseries, timeseries_report, report, **kwargs):
    """structure.py:30-146: initial conditions (educated guess when `extra`, warm-up run when
    kwargs['warm_up'] != 0), then the simulation.  Returns (discharge ndarray [R], gw float) for a 10-vector,
    (discharge [N, R], gw [N]) for an [N, 10] matrix.  Extra keyword: math_mode='fast' | 'literal'."""
    report_type = engine.report_code(report)                   # raises on an unknown report type (:69-70)
    simu_length = len(timeseries) - 1
    delta_sec = delta.total_seconds()
    report_gap = (len(timeseries) - 1) // (len(timeseries_report) - 1)
    n_warm = warm_up_length(kwargs['warm_up'], delta_sec, simu_length) if kwargs['warm_up'] != 0 else 0
    params = np.asarray(nd_parameters, dtype=np.float64)
    single = params.ndim == 1
    forcing = np.stack([np.asarray(nd_rain, dtype=np.float64)[:simu_length],
                        np.asarray(nd_peva, dtype=np.float64)[:simu_length]], axis=1)
    if report_type == engine.REPORT_SUMMARY and (simu_length % report_gap or n_warm % report_gap):
        # what np.reshape raises in the reference (structure.py:190), for the run or for its warm-up
        bad = simu_length if simu_length % report_gap else n_warm
        raise ValueError("cannot reshape array of size {} into shape ({})".format(bad, report_gap))
    out = engine.run_ensemble(params.reshape(-1, 10), forcing, float(area_m2), delta_sec, n_warm, report_gap,
                              report=report, extra=extra if extra else None,
                              math_mode=kwargs.get('math_mode', 'fast'))
    discharge = out.discharge.cpu().numpy()
    gw = out.gw.cpu().numpy()
    if single:
        return np.ascontiguousarray(discharge[0]), float(gw[0])
    return np.ascontiguousarray(discharge), gw


def run_all_steps(area_m2, delta_sec, length_simu, nd_rain, nd_peva, nd_parameters, nd_initial, report_type,
                  report_gap):
    """structure.py:149-197 -> (discharge, groundwater_component, last 19-vector); bit-faithful literal mode."""
    return engine.allsteps(area_m2, delta_sec, length_simu, nd_rain, nd_peva, nd_parameters, nd_initial,
                           report_type, report_gap)


def run_one_step(area_m2, time_delta_sec, c_in_rain, c_in_peva,
                 c_p_t, c_p_c, c_p_h, c_p_d, c_p_s, c_p_z, c_p_sk, c_p_fk, c_p_gk, r_p_rk,
                 c_s_v_h2o_ove, c_s_v_h2o_dra, c_s_v_h2o_int, c_s_v_h2o_sgw, c_s_v_h2o_dgw,
                 c_s_v_h2o_ly1, c_s_v_h2o_ly2, c_s_v_h2o_ly3, c_s_v_h2o_ly4, c_s_v_h2o_ly5, c_s_v_h2o_ly6,
                 r_s_v_riv):
    """structure.py:200-264 -> tuple of 19 floats (outputs then states)."""
    return tuple(engine.onestep(
        area_m2, time_delta_sec, c_in_rain, c_in_peva,
        c_p_t, c_p_c, c_p_h, c_p_d, c_p_s, c_p_z, c_p_sk, c_p_fk, c_p_gk, r_p_rk,
        c_s_v_h2o_ove, c_s_v_h2o_dra, c_s_v_h2o_int, c_s_v_h2o_sgw, c_s_v_h2o_dgw,
        c_s_v_h2o_ly1, c_s_v_h2o_ly2, c_s_v_h2o_ly3, c_s_v_h2o_ly4, c_s_v_h2o_ly5, c_s_v_h2o_ly6,
        r_s_v_riv).tolist())


def run_one_step_catchment(area_m2, time_gap_sec, c_in_rain, c_in_peva,
                           c_p_t, c_p_c, c_p_h, c_p_d, c_p_s, c_p_z, c_p_sk, c_p_fk, c_p_gk,
                           c_s_v_ove, c_s_v_dra, c_s_v_int, c_s_v_sgw, c_s_v_dgw,
                           c_s_v_ly1, c_s_v_ly2, c_s_v_ly3, c_s_v_ly4, c_s_v_ly5, c_s_v_ly6):
    """structure.py:267-458 -> tuple of 17 floats: actual evapotranspiration and the five outflows, then the five
    reservoir and six soil-layer volumes.  The river does not feed back into the catchment, so this is the catchment
    part of one full step on the GPU (river left empty)."""
    v = engine.onestep(area_m2, time_gap_sec, c_in_rain, c_in_peva,
                       c_p_t, c_p_c, c_p_h, c_p_d, c_p_s, c_p_z, c_p_sk, c_p_fk, c_p_gk, 1.0,
                       c_s_v_ove, c_s_v_dra, c_s_v_int, c_s_v_sgw, c_s_v_dgw,
                       c_s_v_ly1, c_s_v_ly2, c_s_v_ly3, c_s_v_ly4, c_s_v_ly5, c_s_v_ly6, 0.0).tolist()
    return tuple(v[0:6] + v[7:18])


def run_one_step_river(time_gap_sec, r_in_q_riv, r_p_rk, r_s_v_riv):
    """structure.py:461-503 -> (r_out_q_riv, r_s_v_riv)."""
    return tuple(engine.river_step_batch([[time_gap_sec, r_in_q_riv, r_p_rk, r_s_v_riv]])[0].tolist())
