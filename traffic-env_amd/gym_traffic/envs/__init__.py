from gym_traffic.envs.traffic_env import TrafficEnv  # noqa: F401
from gym_traffic.envs.roadgraph import GridRoad  # noqa: F401
from gym_traffic.envs.vec_env import TrafficVecEnv  # noqa: F401
