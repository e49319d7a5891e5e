"""Every microkernel family x output mode x arithmetic x solver family compiles
for gfx950 (hipcc, no launch): `tests/compile_matrix.py`.  The translation
units are pre-compiled into the JIT cache by ``__graft_entry__.build()``; on a
cold cache this test compiles them itself (8 hipcc at a time, ~3 minutes)."""
import compile_matrix


def test_every_solver_mode_compiles_for_every_microkernel_family():
    labels, failures = compile_matrix.compile_all()
    fams = {label.split('/')[0] for label in labels}
    assert fams == {'constant', 'kronecker', 'delta_x_sqexp',
                    'molecular_tables', 'rational_quadratic',
                    'normalized_dot_product', 'operators', 'additive',
                    'convolution'}
    modes = {label.split('/')[2] for label in labels}
    assert modes == set(compile_matrix.MODES)
    assert {label.split('/')[1] for label in labels} == {'f32', 'f64'}
    kernels = {label.split('/')[3] for label in labels}
    # solver families: static one-wave, dynamic multi-wave, on-the-fly,
    # two-stage, general -- in both gradient forms and with nodal outputs
    for needle in ('_L16', 'oc8_W4_S32', 'oc8_W16_S40', 'oc0_W4_S0',
                   'mgk_f32_W1_S8', 'mgk_f64_W16_S16', '_general_', '_mfma_', '_stream_',
                   '_ngrad',
                   '_maximin', '_nodal', '_tab', '_C2'):
        assert any(needle in k for k in kernels), needle
    assert len(labels) > 500
    assert not failures, '\n\n'.join(f'{a}:\n{b}' for a, b in failures[:5])
