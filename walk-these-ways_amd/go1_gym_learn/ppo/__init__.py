"""The older teacher-student runner (mirror of reference go1_gym_learn/ppo/__init__.py:44-298): `Runner`,
`RunnerArgs`, `caches`, and the package-level names its modules import from here (`ActorCritic`, `RolloutStorage`).
Same constructor, `learn()` signature, checkpoint / TorchScript export names as the reference; train.py does not use
it (it drives go1_gym_learn.ppo_cse), so it is plain PyTorch on top of the same HIP environment (SURVEY.md §8f
rank 4).  Episode statistics are taken from the env's lazy `train/episode` mapping at log time, as in ppo_cse."""
import copy
import os
import time
from collections import deque

import torch
from params_proto import PrefixProto

from .actor_critic import ActorCritic
from .rollout_storage import RolloutStorage


def class_to_dict(obj) -> dict:
    """nested plain-dict view of a params_proto-style class tree (public attributes; `terrain` skipped)."""
    if not hasattr(obj, "__dict__"):
        return obj
    out = {}
    for key in dir(obj):
        if key.startswith("_") or key == "terrain":
            continue
        val = getattr(obj, key)
        out[key] = [class_to_dict(v) for v in val] if isinstance(val, list) else class_to_dict(val)
    return out


class DataCaches:
    def __init__(self, curriculum_bins):
        from go1_gym_learn.ppo.metrics_caches import DistCache, SlotCache
        self.slot_cache = SlotCache(curriculum_bins)
        self.dist_cache = DistCache()


caches = DataCaches(1)


class RunnerArgs(PrefixProto, cli=False):
    algorithm_class_name = 'PPO'
    num_steps_per_env = 24
    max_iterations = 1500
    save_interval = 400
    save_video_interval = 100
    log_freq = 10
    resume = False
    load_run = -1
    checkpoint = -1
    resume_path = None


class Runner:
    def __init__(self, env, device='cpu'):
        from .ppo import PPO
        self.device = device
        self.env = env
        actor_critic = ActorCritic(env.num_obs, env.num_privileged_obs, env.num_obs_history, env.num_actions).to(device)
        self.alg = PPO(actor_critic, device=device)
        self.num_steps_per_env = RunnerArgs.num_steps_per_env
        self.alg.init_storage(env.num_train_envs, self.num_steps_per_env, [env.num_obs], [env.num_privileged_obs],
                              [env.num_obs_history], [env.num_actions])
        self.tot_timesteps = 0
        self.tot_time = 0
        self.current_learning_iteration = 0
        self.last_recording_it = 0
        self.env.reset()

    def learn(self, num_learning_iterations, init_at_random_ep_len=False, eval_freq=100, eval_expert=False):
        from ml_logger import logger
        assert logger.prefix, "you will overwrite the entire instrument server"
        logger.start('start', 'epoch', 'episode', 'run', 'step')
        env, ac = self.env, self.alg.actor_critic
        if init_at_random_ep_len:
            buf = env.episode_length_buf          # in place: the device buffer is what the simulator reads
            buf.copy_(torch.randint_like(buf, high=int(env.max_episode_length)))
        n = env.num_train_envs
        obs_dict = env.get_observations()
        ac.train()
        rewbuffer, lenbuffer = deque(maxlen=100), deque(maxlen=100)
        if hasattr(env, "curriculum"):
            caches.__init__(curriculum_bins=len(env.curriculum))
        elif hasattr(env, "curricula"):
            # train.py's env keeps one curriculum per gait category; PPO.update logs the adaptation residual per bin
            # through a uint8 slot index (reference ppo.py:150), i.e. modulo 256
            caches.__init__(curriculum_bins=max(256, max(len(c) for c in env.curricula)))
        infos = {}
        it = self.current_learning_iteration
        for it in range(self.current_learning_iteration, self.current_learning_iteration + num_learning_iterations):
            with torch.inference_mode():
                for _ in range(self.num_steps_per_env):
                    obs, priv, hist = (obs_dict[k].to(self.device) for k in ("obs", "privileged_obs", "obs_history"))
                    actions = self.alg.act(obs[:n], priv[:n], hist[:n])
                    if env.num_envs > n:          # evaluation environments ride along with the teacher or the student
                        extra = ac.act_teacher(obs[n:], priv[n:]) if eval_expert else ac.act_student(obs[n:], hist[n:])
                        actions = torch.cat((actions, extra), dim=0)
                    obs_dict, rewards, dones, infos = env.step(actions)
                    self.alg.process_env_step(rewards[:n].to(self.device), dones[:n].to(self.device), infos)
                self.alg.compute_returns(obs_dict["obs"][:n].to(self.device), obs_dict["privileged_obs"][:n].to(self.device))
                if it % eval_freq == 0:
                    if hasattr(env, "reset_evaluation_envs"):
                        env.reset_evaluation_envs()
                    logger.save_pkl({"iteration": it, **caches.slot_cache.get_summary(), **caches.dist_cache.get_summary()},
                                    path="curriculum/info.pkl", append=True)
            mean_value_loss, mean_surrogate_loss, mean_adaptation_module_loss = self.alg.update()
            logger.store_metrics(time_elapsed=logger.since('start'), time_iter=logger.split('epoch'),
                                 adaptation_loss=mean_adaptation_module_loss, mean_value_loss=mean_value_loss,
                                 mean_surrogate_loss=mean_surrogate_loss)
            if RunnerArgs.save_video_interval:
                self.log_video(it)
            self.tot_timesteps += self.num_steps_per_env * env.num_envs
            if logger.every(RunnerArgs.log_freq, "iteration", start_on=1):
                stats = infos.get('train/episode') if hasattr(infos, "get") else None
                if stats is not None:
                    with logger.Prefix(metrics="train/episode"):
                        logger.store_metrics(**(stats.consume() if hasattr(stats, "consume") else stats))
                logger.log_metrics_summary(key_values={"timesteps": self.tot_timesteps, "iterations": it})
                logger.job_running()
            if it % RunnerArgs.save_interval == 0:
                self.save(it)
        self.current_learning_iteration += num_learning_iterations
        self.save(it)
        return rewbuffer, lenbuffer

    def save(self, it):
        """checkpoint + TorchScript export with the reference's file names (:213-231, :235-254)."""
        from ml_logger import logger
        ac = self.alg.actor_critic
        with logger.Sync():
            logger.torch_save(ac.state_dict(), f"checkpoints/ac_weights_{it:06d}.pt")
            logger.duplicate(f"checkpoints/ac_weights_{it:06d}.pt", "checkpoints/ac_weights_last.pt")
            path = './tmp/legged_data'
            os.makedirs(path, exist_ok=True)
            for name, module in (("adaptation_module_latest.jit", ac.adaptation_module), ("body_latest.jit", ac.actor_body)):
                torch.jit.script(copy.deepcopy(module).to('cpu')).save(f'{path}/{name}')
                logger.upload_file(file_path=f'{path}/{name}', target_path="checkpoints/", once=False)

    def log_video(self, it):
        from ml_logger import logger
        env = self.env
        if it - self.last_recording_it >= RunnerArgs.save_video_interval:
            env.start_recording()
            if env.num_eval_envs > 0:
                env.start_recording_eval()
            self.last_recording_it = it
        frames = env.get_complete_frames()
        if len(frames) > 0:
            env.pause_recording()
            logger.save_video(frames, f"videos/{it:05d}.mp4", fps=1 / env.dt)
        if env.num_eval_envs > 0:
            frames = env.get_complete_frames_eval()
            if len(frames) > 0:
                env.pause_recording_eval()
                logger.save_video(frames, f"videos/{it:05d}_eval.mp4", fps=1 / env.dt)

    def get_inference_policy(self, device=None):
        self.alg.actor_critic.eval()
        if device is not None:
            self.alg.actor_critic.to(device)
        return self.alg.actor_critic.act_inference

    def get_expert_policy(self, device=None):
        self.alg.actor_critic.eval()
        if device is not None:
            self.alg.actor_critic.to(device)
        return self.alg.actor_critic.act_expert
