#!/usr/bin/env python3
"""Multi-GPU counterpart of `python3 airfoil_dqn.py` (reference airfoil_dqn.py:343-514):

    python train.py --config configs/ray_ys930.yaml                                   # 1 GPU
    torchrun --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 train.py --config ...   # 8 GPUs (RCCL)

One process per GPU; every rank steps `--envs` environments (VecEnv2DAirfoil), all ranks apply the same
all-reduced gradient, replay transitions are optionally all-gathered.  The yaml is the reference's own format
(`flow_config`, `agent_params`, `optimizer`, `epsilon`); `geometry_params.mesh` may point at an .xdmf or .npz file."""
import argparse
import os as _os

# single-threaded BLAS / OpenMP pools BEFORE numpy / scipy / torch are imported: the hosts expose hundreds of logical CPUs
# under a small CPU quota, and idle pool threads that keep spinning starve the threads that launch kernels
for _k in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
    _os.environ.setdefault(_k, "1")
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")   # concurrent streams on separate hardware queues (meshdqn_amd/__init__.py)
import os

import numpy as np
import yaml


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", required=True)
    ap.add_argument("--envs", type=int, default=128, help="environments per GPU")
    ap.add_argument("--steps", type=int, default=1000, help="batched rollout steps")
    ap.add_argument("--flow-steps", type=int, default=0, help="IPCS steps on the coarsened mesh per env step (S3)")
    ap.add_argument("--share-replay", action="store_true")
    ap.add_argument("--save-dir", default="training_results/run")
    ap.add_argument("--restart", action="store_true",
                    help="continue the run found in --save-dir: both Q-networks, optimiser / scheduler state, epsilon "
                         "counters and the logs (files of restart n carry n 'restart_' / 'RESTART_' prefixes, like the "
                         "reference's RESTART_NUM chain, airfoil_dqn.py:359-366)")
    ap.add_argument("--save-every", type=int, default=50, help="batched steps between checkpoints / log writes (0 = only at the end)")
    ap.add_argument("--host-loop", action="store_true",
                    help="host-driven loop (train_loop_vec: autograd replayed as a HIP graph, one read-back per step) instead "
                         "of the device-resident one (train_loop_device: replay, sampling, forward + backward and Adam as kernels)")
    args = ap.parse_args()
    from meshdqn_amd.env import Env2DAirfoil
    from meshdqn_amd.trainer import DistContext, DQNTrainer, TrainingLog, train_loop_device, train_loop_vec
    from meshdqn_amd.vec_env import VecEnv2DAirfoil
    import torch
    torch.set_num_threads(1)   # (many-core hosts under a CPU quota: the CPU-side tensor ops are tiny, no intra-op pool)
    cfg = yaml.safe_load(open(args.config))
    ctx = DistContext()
    opt = cfg.get("optimizer", {})          # reference yaml sections: optimizer / epsilon (configs/ray_ys930.yaml)
    eps = cfg.get("epsilon", {})
    ap_ = cfg["agent_params"]
    trainer = DQNTrainer(n_actions=int(ap_["N_closest"]), num_inputs=2 + 3 * (int(ap_["solver_steps"]) // int(ap_["save_steps"])),
                         ctx=ctx, lr=float(opt.get("lr", 1e-5)), weight_decay=float(opt.get("weight_decay", 1e-6)),
                         batch_size=int(opt.get("batch_size", 32)), gamma=float(eps.get("gamma", 1.0)),
                         target_update=int(ap_.get("target_update", 50)))
    # restart chain: restart n reads the checkpoint with n - 1 "restart_" prefixes and writes with n (every rank loads the
    # same files, so the replicas stay identical)
    restart_num, steps_done0 = 0, None
    if args.restart:
        restart_num = sum(f.endswith("policy_net_1.pt") for f in os.listdir(args.save_dir))
        if restart_num == 0:
            raise SystemExit(f"--restart: no policy_net_1.pt checkpoint in {args.save_dir}")
        extra = trainer.load(args.save_dir, "restart_" * (restart_num - 1))
        steps_done0 = extra.get("steps_done")
        if steps_done0 is not None and len(steps_done0) != args.envs:      # other batch size: restart the counters at the mean
            steps_done0 = np.full(args.envs, int(np.mean(steps_done0)), np.int64)
    prefix = "restart_" * restart_num
    base = Env2DAirfoil(cfg, compute_device=ctx.device)          # ground truth + snapshots (the reference's first reset())
    venv = VecEnv2DAirfoil(cfg, args.envs, compute_device=ctx.device, base_env=base, flow_steps=args.flow_steps,
                           flow_overlap=args.flow_steps > 0)
    log = TrainingLog(args.save_dir, restart=args.restart, restart_num=restart_num) if ctx.rank == 0 else None

    def checkpoint(step, steps_done):
        if ctx.rank == 0:                                        # (the reference writes after every episode)
            trainer.save(args.save_dir, prefix, extra=dict(steps_done=np.asarray(steps_done).copy(), batched_steps=step))
            log.write()                                          # reward / rewards / losses / actions / eps .npy

    device_loop = ctx.device.type == "cuda" and venv.gpu_remesh and not args.host_loop
    loop = train_loop_device if device_loop else train_loop_vec
    out = loop(trainer, venv, args.steps, log=log, eps_decay=float(eps.get("decay", 10000)),
               eps_start=float(eps.get("start", 1.0)), eps_end=float(eps.get("end", 0.01)),
               share_replay=args.share_replay, steps_done0=steps_done0, every=args.save_every, on_every=checkpoint)
    if ctx.rank == 0:
        os.makedirs(args.save_dir, exist_ok=True)
        checkpoint(args.steps, out["steps_done"])
        np.save(os.path.join(args.save_dir, prefix + "step_rewards.npy"), out["rewards"])
        yaml.safe_dump(cfg, open(os.path.join(args.save_dir, "config.yaml"), "w"))
        print(f"ranks {ctx.world}: {args.steps} batched steps x {args.envs} envs/rank, mean reward {out['rewards'].mean():.4f}")
    ctx.close()


if __name__ == "__main__":
    main()
