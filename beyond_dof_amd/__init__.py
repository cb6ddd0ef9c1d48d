"""beyond_dof_amd — MI355X-native multislice Fresnel forward + adjoint + Adam engine that drops in for
the forward+gradient loop of mdw771/beyond_dof's cnn_propagator."""
__version__ = '0.1.0'
