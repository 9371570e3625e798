"""Algorithm parameters of the hot path, values of the reference's
remixt/defaults.py:113-163, with its `get_param` overlay semantics
(remixt/config.py:5-12: a key present in the config dict wins)."""

max_copy_number = 12
tumour_mix_fractions = [0.45, 0.3, 0.2, 0.1]
min_ploidy = 1.5
max_ploidy = 6.0
h_normal = None
h_tumour = None
max_prop_diverge = 0.5
normal_contamination = True
likelihood_min_segment_length = 10000
likelihood_min_proportion_genotyped = 0.01
divergence_weights = [1e-6, 1e-7, 1e-8]
num_em_iter = 5
num_update_iter = 5
disable_breakpoints = False
do_h_update = True
is_female = True


def get_param(config, name):
    if config is not None and name in config:
        return config[name]
    return globals()[name]


def get_sample_config(config, sample_id):
    """remixt/config.py:56-59: the config with the `sample_specific[sample_id]` overrides applied on top."""
    sample_config = dict(config or {})
    sample_config.update((config or {}).get('sample_specific', dict()).get(sample_id, dict()))
    return sample_config
