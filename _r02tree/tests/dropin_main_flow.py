"""A caller written against the REFERENCE's module paths, following the steps of its main.py:7-44 (arguments -> env from
smac.env -> env_info into args -> Logger -> Runner -> run / evaluate) with a short horizon.  Run through the launcher
(`python -m marl_amd.dropin tests/dropin_main_flow.py`) it must resolve every import to marl_amd and train on the GPU;
tests/test_gpu_runner.py::test_reference_style_main_flow_on_dropin does exactly that."""
import sys

from runner import Runner
from smac.env import StarCraft2Env
from common.arguments import get_common_args, get_mixer_args, get_RTW_args
from utils.logging import Logger, get_logger

if __name__ == '__main__':
    args = get_common_args()
    get_mixer_args(args)
    get_RTW_args(args)
    args.n_steps, args.evaluate_cycle, args.evaluate_epoch = 4000, 2000, 0
    env = StarCraft2Env(map_name=args.map, step_mul=args.step_mul, difficulty=args.difficulty,
                        game_version=args.game_version, replay_dir=args.replay_dir)
    env_info = env.get_env_info()
    args.n_actions, args.n_agents = env_info["n_actions"], env_info["n_agents"]
    args.state_shape, args.obs_shape, args.episode_limit = env_info["state_shape"], env_info["obs_shape"], env_info["episode_limit"]
    args.batch_size = max(args.batch_size, env.n_envs)
    args.buffer_size = 4 * env.n_envs
    log = Logger()
    get_logger().info("reference-style main flow on %s", type(env).__module__)
    runner = Runner(env, log, args)
    loss = runner.run(0)
    env.close()
    mods = {type(runner).__module__, type(runner.learner).__module__, type(runner.mac).__module__, type(env).__module__}
    print("MAIN_FLOW_OK loss=%.5f train_steps=%d modules=%s" % (loss, runner.train_steps, sorted(mods)))
    sys.exit(0 if all(m.startswith("marl_amd") for m in mods) else 1)
