classdef HipSim < handle
	%HipSim  B independent rollouts of one redmax.Scene on one or several MI355X (libredmax_hip.so through redmax_hip_mex).
	%
	%   sim = redmax.HipSim(scene, batch, devices)  scene: an initialised redmax.Scene (or a desc struct); devices: a device
	%                                               index or a VECTOR of them, e.g. 0:7 - the batch is split into one contiguous
	%                                               shard per entry, every array below stays the whole batch
	%   sim.setState(q, qdot)                       nr x batch, the DOF order of Joint.getQ
	%   [T,V,stats,Q,Qdot] = sim.step(itype, h, nsteps, opts)   itype 1: BDF1, 2: SDIRK2 + BDF2; all devices step concurrently
	%   sim.stepAsync(itype, h, nsteps, opts, record);  ...  [T,V,stats,Q,Qdot] = sim.sync();   the same, MATLAB free in between
	%   [q, qdot] = sim.getState()
	%   delete(sim)
	%
	% Replaces the interpreter-side simLoop / newton / evalBDF1 / computeValues of matlab-diff/driverRedMaxBDF1.m:57-243
	% (and driverRedMaxBDF2.m) for a whole batch of trajectories; the Scene/Joint/Body classes are the reference's.

	properties (SetAccess = private)
		h      % uint64 handle of the gateway
		nr     % reduced DOFs
		nm     % maximal DOFs
		nsph   % spherical joints (Euler charts live on the device, see getCharts)
		batch  % trajectories
		idxR   % 0-based reduced index of every listed joint's first DOF (-1: fixed)
		devices      % device of every shard
		shardFirst   % 0-based index of every shard's first trajectory
		shardCount   % trajectories per shard
	end

	methods
		function this = HipSim(scene, batch, devices)
			if nargin < 2, batch = 1; end
			if nargin < 3, devices = 0; end
			if isstruct(scene)
				desc = scene;
			else
				desc = redmax.flattenScene(scene);
			end
			this.h = redmax_hip_mex('create', desc, batch, double(devices(:)'));
			info = redmax_hip_mex('info', this.h);
			this.nr = info.nr; this.nm = info.nm; this.nsph = info.nsph; this.batch = info.batch; this.idxR = info.idxR;
			this.devices = info.devices; this.shardFirst = info.shard_first; this.shardCount = info.shard_count;
		end

		function delete(this)
			if ~isempty(this.h)
				redmax_hip_mex('destroy', this.h);
				this.h = [];
			end
		end

		function setState(this, q, qdot)
			% Joint.setQ for the batch; a single column is replicated over the batch
			if size(q,2) == 1 && this.batch > 1
				q = repmat(q, 1, this.batch); qdot = repmat(qdot, 1, this.batch);
			end
			redmax_hip_mex('set', this.h, q, qdot);
		end

		function [q, qdot] = getState(this)
			[q, qdot] = redmax_hip_mex('get', this.h);
		end

		function varargout = step(this, itype, hstep, nsteps, opts)
			if nargin < 5, opts = struct(); end
			[varargout{1:max(nargout,1)}] = redmax_hip_mex('step', this.h, itype, hstep, nsteps, opts);
		end

		function stepAsync(this, itype, hstep, nsteps, opts, record)
			% launch simLoop on every device and return; record: 1 = T, V (default), +2 = Q, Qdot, +4 = charts
			if nargin < 5 || isempty(opts), opts = struct(); end
			if nargin < 6, record = 1; end
			redmax_hip_mex('step_async', this.h, itype, hstep, nsteps, opts, record);
		end

		function varargout = sync(this)
			% wait for the launches of stepAsync and gather: [T, V, stats, Q, Qdot, C] as step
			[varargout{1:max(nargout,1)}] = redmax_hip_mex('sync', this.h);
		end

		function [wall, kernel, t0, t1] = timing(this)
			% wall clock ms of the last step and, per shard, kernel ms and launch start / end (rmx_group_timing)
			[wall, kernel, t0, t1] = redmax_hip_mex('timing', this.h);
		end

		function t = ticks(this)
			% shader-clock ticks every rollout's wavefront spent in the last step launch (rmx_step_ticks)
			t = redmax_hip_mex('ticks', this.h);
		end

		function [T, V] = euler(this, hstep, nsteps)
			[T, V] = redmax_hip_mex('euler', this.h, hstep, nsteps);
		end

		function varargout = evalResidual(this, q, qA, qB, eta)
			% g (and H when two outputs are requested): evalBDF1 is evalResidual(q1, q0, q0 + h*qdot0, h)
			[varargout{1:max(nargout,1)}] = redmax_hip_mex('eval', this.h, q, qA, qB, eta);
		end

		function varargout = computeValues(this, q, qdot, varargin)
			% [M,f,K,D,dMv] = computeValues(q, qdot [, v]): the full output of the reference's computeValues (driverRedMaxBDF1.m:188-243)
			% at (q, qdot), per rollout; dMv(:,i,b) = dMdq(:,:,i) * v(:,b)
			[varargout{1:max(nargout,1)}] = redmax_hip_mex('values', this.h, q, qdot, varargin{:});
		end

		function [T, V] = energy(this)
			[T, V] = redmax_hip_mex('energy', this.h);
		end

		function [q, qdot, path] = gather(this, root)
			% the final gather of the sharded batch with device-resident destinations (rmx_group_gather: RCCL over the group's devices):
			% every shard's device (root omitted) or shard `root`'s device alone (0-based) ends up with the whole batch; q, qdot (nr x batch):
			% that copy read back; path: 'rccl:allgather' | 'rccl:broadcast' | 'rccl:sendrecv' | 'copy'
			if nargin < 2
				root = -1;
			end
			[q, qdot, path] = redmax_hip_mex('gather', this.h, root);
		end

		function c = getCharts(this)
			c = redmax_hip_mex('getcharts', this.h);
		end

		function setCharts(this, c)
			redmax_hip_mex('setcharts', this.h, int32(c));
		end

		function [P, dPdp, stats] = adjoint(this, hstep, nsteps, task, p, itype)
			% taskObjective of driverRedMaxAdjointBDF1.m (itype 1, default) / driverRedMaxAdjointBDF2.m (itype 2)
			if nargin < 6
				itype = 1;
			end
			[P, dPdp, stats] = redmax_hip_mex('adjoint', this.h, hstep, nsteps, task, p, itype);
		end
	end
end
