"""`Runner`: rollout / learn loop, logging and checkpoint export (mirror of reference
go1_gym_learn/ppo_cse/__init__.py:44-308).  Same constructor, `learn()` signature, RunnerArgs, checkpoint file
names (`checkpoints/ac_weights_{it:06d}.pt`, `ac_weights_last.pt`, `adaptation_module_latest.jit`,
`body_latest.jit`).  Episode statistics are read from the env's lazy `train/episode` mapping at log time instead
of being pushed to the logger every step (which would force ~20 tiny reductions and host reads per step)."""
import copy
import os
import time
from collections import deque  # noqa: F401

import torch
from ml_logger import logger
from params_proto import PrefixProto

from .actor_critic import ActorCritic
from .rollout_storage import RolloutStorage


def class_to_dict(obj) -> dict:
    """nested plain-dict view of a params_proto-style class tree: public attributes, `terrain` skipped (reference :14-29)"""
    if not hasattr(obj, "__dict__"):
        return obj
    view = {}
    for key in (k for k in dir(obj) if not k.startswith("_") and k != "terrain"):
        val = getattr(obj, key)
        view[key] = [class_to_dict(v) for v in val] if isinstance(val, list) else class_to_dict(val)
    return view


class DataCaches:
    def __init__(self, curriculum_bins):
        from go1_gym_learn.ppo.metrics_caches import DistCache, SlotCache
        self.slot_cache = SlotCache(curriculum_bins)
        self.dist_cache = DistCache()


caches = DataCaches(1)


class RunnerArgs(PrefixProto, cli=False):
    algorithm_class_name = 'RMA'
    num_steps_per_env = 24
    max_iterations = 1500
    save_interval = 400
    save_video_interval = 100
    log_freq = 10
    resume = False
    load_run = -1
    checkpoint = -1
    resume_path = None
    resume_curriculum = True


class Runner:
    def __init__(self, env, device='cpu'):
        from .ppo import PPO
        self.device = device
        self.env = env
        actor_critic = ActorCritic(self.env.num_obs, self.env.num_privileged_obs, self.env.num_obs_history,
                                   self.env.num_actions).to(self.device)
        if RunnerArgs.resume:
            # the reference pulls from a hard-coded remote ml_logger server (:76-91); here: a local run directory
            from ml_logger import ML_Logger
            loader = ML_Logger()
            loader.configure(prefix=RunnerArgs.resume_path, root=logger.root)
            actor_critic.load_state_dict(loader.load_torch("checkpoints/ac_weights_last.pt", map_location=self.device))
            if hasattr(self.env, "curricula") and RunnerArgs.resume_curriculum:
                last = loader.load_pkl("curriculum/distribution.pkl")[-1]["distribution"]
                for gait_id, gait_name in enumerate(self.env.category_names):
                    self.env.curricula[gait_id].weights = last[f"weights_{gait_name}"]
                self.env.sync_curricula_to_device()
        self.alg = PPO(actor_critic, device=self.device)
        self.num_steps_per_env = RunnerArgs.num_steps_per_env
        # HistoryWrapper: obs_history is a sliding window over obs that nothing rewrites during a rollout -> ring storage
        self.alg.init_storage(self.env.num_train_envs, self.num_steps_per_env, [self.env.num_obs],
                              [self.env.num_privileged_obs], [self.env.num_obs_history], [self.env.num_actions],
                              sliding_history=hasattr(self.env, "obs_history_length"))
        self.tot_timesteps = 0
        self.tot_time = 0
        self.current_learning_iteration = 0
        self.last_recording_it = 0
        self.collection_time = 0.0
        self.learn_time = 0.0
        self._it_now = 0
        self._eval_module_it = None          # iteration whose weights the nn.Module (evaluation actions) was last refreshed with
        self.env.reset()

    # one policy step of the rollout (reference :139-154)
    def _rollout_step(self, obs_dict, eval_expert=False):
        n = self.env.num_train_envs
        obs, priv, hist = obs_dict["obs"], obs_dict["privileged_obs"], obs_dict["obs_history"]
        actions = self.alg.act(obs[:n], priv[:n], hist[:n])
        if self.env.num_eval_envs > 0:          # evaluation environments: deterministic student / teacher actions (:142-147)
            ac = self.alg.sync_module() if self._eval_module_it != self._it_now else self.alg.actor_critic
            self._eval_module_it = self._it_now
            if eval_expert:
                actions_eval = ac.act_teacher(hist[n:], priv[n:])
            else:
                actions_eval = ac.act_student(hist[n:])
            actions = torch.cat((actions, actions_eval), dim=0)
        obs_dict, rewards, dones, infos = self.env.step(actions)
        self.alg.process_env_step(rewards[:n], dones[:n], infos)
        return obs_dict, infos

    def learn(self, num_learning_iterations, init_at_random_ep_len=False, eval_freq=100, curriculum_dump_freq=500,
              eval_expert=False):
        assert logger.prefix, "you will overwrite the entire instrument server"
        logger.start('start', 'epoch', 'episode', 'run', 'step')
        if init_at_random_ep_len:
            # the reference assigns on the wrapper, which never reaches the base env;
            # the intent (de-phased episodes) is applied to the real buffer here
            buf = self.env.episode_length_buf
            buf.copy_(torch.randint_like(buf, high=int(self.env.max_episode_length)))
        obs_dict = self.env.get_observations()
        self.alg.actor_critic.train()
        infos = {}
        tot_iter = self.current_learning_iteration + num_learning_iterations
        for it in range(self.current_learning_iteration, tot_iter):
            start = time.time()
            with torch.inference_mode():
                self._it_now = it
                for _ in range(self.num_steps_per_env):
                    obs_dict, infos = self._rollout_step(obs_dict, eval_expert)
                stop = time.time()
                self.collection_time = stop - start
                start = stop
                n = self.env.num_train_envs
                self.alg.compute_returns(obs_dict["obs_history"][:n], obs_dict["privileged_obs"][:n])
                if it % curriculum_dump_freq == 0:
                    logger.save_pkl({"iteration": it, **caches.slot_cache.get_summary(), **caches.dist_cache.get_summary()},
                                    path="curriculum/info.pkl", append=True)
                    if 'curriculum/distribution' in infos:
                        logger.save_pkl({"iteration": it, "distribution": dict(infos['curriculum/distribution'])},
                                        path="curriculum/distribution.pkl", append=True)
            (mean_value_loss, mean_surrogate_loss, mean_adaptation_module_loss, mean_decoder_loss, mean_decoder_loss_student,
             mean_adaptation_module_test_loss, mean_decoder_test_loss, mean_decoder_test_loss_student) = self.alg.update()
            self.learn_time = time.time() - start
            logger.store_metrics(time_elapsed=logger.since('start'), time_iter=logger.split('epoch'),
                                 adaptation_loss=mean_adaptation_module_loss, mean_value_loss=mean_value_loss,
                                 mean_surrogate_loss=mean_surrogate_loss, mean_decoder_loss=mean_decoder_loss,
                                 mean_decoder_loss_student=mean_decoder_loss_student,
                                 mean_decoder_test_loss=mean_decoder_test_loss,
                                 mean_decoder_test_loss_student=mean_decoder_test_loss_student,
                                 mean_adaptation_module_test_loss=mean_adaptation_module_test_loss)
            if RunnerArgs.save_video_interval:
                self.log_video(it)
            self.tot_timesteps += self.num_steps_per_env * self.env.num_envs
            if logger.every(RunnerArgs.log_freq, "iteration", start_on=1):
                stats = infos.get('train/episode') if hasattr(infos, "get") else None
                if stats is not None:
                    with logger.Prefix(metrics="train/episode"):
                        logger.store_metrics(**(stats.consume() if hasattr(stats, "consume") else stats))
                faults = infos.get('sim_faults') if hasattr(infos, "get") else None
                if faults is not None and hasattr(faults, "consume"):       # containments of failed environments (go1sim.h Go1FaultBit)
                    fc = faults.consume()
                    fc.update({f"contact_dropped_{k}": v for k, v in fc.pop("contact_dropped_by_class", {}).items()})
                    with logger.Prefix(metrics="sim_faults"):
                        logger.store_metrics(**fc)
                logger.log_metrics_summary(key_values={"timesteps": self.tot_timesteps, "iterations": it})
                logger.job_running()
            if it % RunnerArgs.save_interval == 0:
                self.save(it)
        self.current_learning_iteration += num_learning_iterations
        self.save(it)

    def save(self, it):
        """reference :231-251 / :255-274."""
        self.alg.sync_module()
        with logger.Sync():
            logger.torch_save(self.alg.actor_critic.state_dict(), f"checkpoints/ac_weights_{it:06d}.pt")
            logger.duplicate(f"checkpoints/ac_weights_{it:06d}.pt", "checkpoints/ac_weights_last.pt")
            path = './tmp/legged_data'
            os.makedirs(path, exist_ok=True)
            for name, module in (("adaptation_module_latest.jit", self.alg.actor_critic.adaptation_module),
                                 ("body_latest.jit", self.alg.actor_critic.actor_body)):
                scripted = torch.jit.script(copy.deepcopy(module).to('cpu'))
                scripted.save(f'{path}/{name}')
                logger.upload_file(file_path=f'{path}/{name}', target_path="checkpoints/", once=False)

    def log_video(self, it):
        if it - self.last_recording_it >= RunnerArgs.save_video_interval:
            self.env.start_recording()
            if self.env.num_eval_envs > 0:
                self.env.start_recording_eval()
            self.last_recording_it = it
        frames = self.env.get_complete_frames()
        if len(frames) > 0:
            self.env.pause_recording()
            logger.save_video(frames, f"videos/{it:05d}.mp4", fps=1 / self.env.dt)

    def get_inference_policy(self, device=None):
        self.alg.sync_module()
        self.alg.actor_critic.eval()
        if device is not None:
            self.alg.actor_critic.to(device)
        return self.alg.actor_critic.act_inference

    def get_expert_policy(self, device=None):
        self.alg.sync_module()
        self.alg.actor_critic.eval()
        if device is not None:
            self.alg.actor_critic.to(device)
        return self.alg.actor_critic.act_expert
