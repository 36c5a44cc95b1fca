from ebcsim.state import FullState, JointState, ObservableState  # noqa: F401  (simulator/utils/state.py)
