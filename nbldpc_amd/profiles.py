"""Profile text generator for the reference's positional `NBLDPC.Profile.txt` grammar.

The reference parser (Simulation.cpp:58-107) identifies each value only by the number of label tokens
in front of it, so the label words below are free text of our own choosing; the token counts are what
matter.  Used by tests/golden/make_golden.py (build container), the tests and bench.py (to write profiles into work dirs).
"""

DEFAULTS = dict(
    gfq=256, code="code.txt", puncture_degree=0, method=2, max_iter=50, parallel=1,
    crc_len=8, crc_correct=0, osd_order=-1, osd_factor=0, osd_flag=0,
    ems_nm=32, ems_nc=3, ems_factor=1.0, ems_offset=0.0,
    tems_nr=2, tems_nc=3, tems_factor=1.0, tems_offset=0.0,
    bs_nm=4, bs_nc=2, bs_factor=1.0, bs_offset=0.0,
    snr_begin=2.0, snr_step=1.0, snr_stop=2.0,
    nqam=2, constellation="BPSK.txt", random_msg=1,
    min_err_frame=-1, min_uerr_frame=-1, min_sim_cycle=0, seed=173, show_step=1000000,
)


def profile_text(**kw):
    p = dict(DEFAULTS)
    unknown = set(kw) - set(p)
    if unknown:
        raise KeyError(f"unknown profile keys: {sorted(unknown)}")
    p.update(kw)
    return (
        f"GFq: {p['gfq']}\n"
        f"NB File: {p['code']}\n"
        f"Puncture Degree: {p['puncture_degree']}\n"
        f"Decode Method: {p['method']}\n"
        f"Max Iter: {p['max_iter']}\n"
        f"Parallel: {p['parallel']}\n"
        f"crcLen: {p['crc_len']}\n"
        f"crcCorrect: {p['crc_correct']}\n"
        f"OSD_order: {p['osd_order']}\n"
        f"OSD_factor: {p['osd_factor']}\n"
        f"OSD_flag: {p['osd_flag']}\n"
        f"EMS Nm: {p['ems_nm']}\n"
        f"EMS Nc: {p['ems_nc']}\n"
        f"EMS Factor: {p['ems_factor']}\n"
        f"EMS Offset: {p['ems_offset']}\n"
        f"TEMS Nr: {p['tems_nr']}\n"
        f"TEMS Nc: {p['tems_nc']}\n"
        f"TEMS Factor: {p['tems_factor']}\n"
        f"TEMS Offset: {p['tems_offset']}\n"
        f"BSTEMS Nm: {p['bs_nm']}\n"
        f"BSTEMS Nc: {p['bs_nc']}\n"
        f"BSTEMS Factor: {p['bs_factor']}\n"
        f"BSTEMS Offset: {p['bs_offset']}\n"
        f"SNR Begin: {p['snr_begin']}\n"
        f"SNR Step: {p['snr_step']}\n"
        f"SNR Stop: {p['snr_stop']}\n"
        f"nQAM: {p['nqam']}\n"
        f"Constellation: {p['constellation']}\n"
        f"Random Msg: {p['random_msg']}\n"
        f"Min Err Frame: {p['min_err_frame']}\n"
        f"Min UErr Frame: {p['min_uerr_frame']}\n"
        f"Min Sim Cycle: {p['min_sim_cycle']}\n"
        f"Random Seed: {p['seed']}\n"
        f"Show Frame Step: {p['show_step']}\n"
    )
