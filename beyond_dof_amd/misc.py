"""summary.txt writer of the reconstruction entry points (cnn_propagator/misc.py:61-77)."""
import os

SUMMARY_PRESET_PTYCHO = ['obj_size', 'probe_size', 'output_folder', 'theta_downsample', 'n_theta', 'n_pos', 'n_epochs',
                         'learning_rate', 'alpha_d', 'alpha_b', 'gamma', 'n_dp_batch', 'minibatch_size', 'free_prop_cm',
                         'psize_cm', 'energy_ev', 'fname', 'probe_mag_sigma', 'probe_phase_sigma', 'probe_phase_max',
                         # not in the reference's preset: the precision the adjoint sweep was ASKED to run in and the one it runs in
                         'adjoint_precision', 'adjoint_precision_effective']
SUMMARY_PRESET_FF = ['obj_size', 'output_folder', 'theta_downsample', 'n_theta', 'n_epochs', 'learning_rate', 'alpha_d',
                     'alpha_b', 'gamma', 'minibatch_size', 'free_prop_cm', 'psize_cm', 'energy_ev', 'fname', 'object_type']


def create_summary(save_path, locals_dict, var_list=None, preset=None):
    """One `name  value` line per variable of the preset; variables the caller does not have are written as None."""
    if preset == 'ptycho':
        var_list = SUMMARY_PRESET_PTYCHO
    elif preset == 'fullfield':
        var_list = SUMMARY_PRESET_FF
    if not os.path.exists(save_path):
        os.makedirs(save_path)
    with open(os.path.join(save_path, 'summary.txt'), 'w') as f:
        for var_name in var_list:
            # the reference's '{:<20}{}' layout; a name that fills the column still gets a blank before its value
            f.write('{:<20}{}{}\n'.format(var_name, '' if len(var_name) < 20 else ' ', str(locals_dict.get(var_name))))
