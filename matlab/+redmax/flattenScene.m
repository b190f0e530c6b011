function desc = flattenScene(scene)
%flattenScene  Arrays of rmx_model_desc (include/redmax_hip.h) for an initialised redmax.Scene.
%
%   desc = redmax.flattenScene(scene)      after scene.init()
%
% This is the only place the MEX path reads the reference's handle-object graph (Scene.joints / Scene.bodies /
% Scene.forces as Scene.init() leaves them, matlab-diff/+redmax/Scene.m:59-119).  Joints stay in the scene's listing
% order; the library numbers the DOFs leaf-to-root exactly as Joint.countDofs does (Joint.m:149-158), so reduced
% vectors exchanged with redmax_hip_mex are the ones Joint.getQ / Joint.setQ use.
%
% This file adds a function to the +redmax package; it does not replace any file of the reference.  Put this
% repository's matlab/ directory on the MATLAB path next to the reference's matlab-diff/ (package folders merge).

n = numel(scene.joints);
if numel(scene.bodies) ~= n
	error('redmax:hip','flattenScene: every joint needs exactly one body');
end
desc.njoints   = n;
desc.parent    = zeros(1,n,'int32');
desc.type      = zeros(1,n,'int32');
desc.axis      = repmat([0;0;1],1,n);
desc.E0_pj     = repmat(eye(4),1,1,n);
desc.E0_ji     = repmat(eye(4),1,1,n);
desc.I_i       = zeros(6,n);
desc.qRest     = zeros(1,n);
desc.tau       = zeros(1,n);
desc.stiffness = zeros(1,n);
desc.damping   = zeros(1,n);
desc.qLimL     = zeros(1,n);
desc.qLimU     = zeros(1,n);
desc.qLimK     = zeros(1,n);
desc.qLimD     = zeros(1,n);
desc.plane     = repmat([1;0;0;0;1;0],1,n);
desc.grav      = scene.grav(:);
nr = 0;
for i = 1 : n
	nr = nr + scene.joints{i}.ndof;
end
desc.qRestR = zeros(nr,1);

% RMX_JOINT_* of include/redmax_hip.h, by class
types = {'redmax.JointFixed',0; 'redmax.JointRevolute',1; 'redmax.JointPrismatic',2; 'redmax.JointPlanar',3; ...
	'redmax.JointTranslational',4; 'redmax.JointUniversal',5; 'redmax.JointFree2D',6; 'redmax.JointSpherical',7; ...
	'redmax.JointFree3D',8};

for i = 1 : n
	j = scene.joints{i};
	if scene.bodies{i} ~= j.body
		error('redmax:hip','flattenScene: bodies must be listed in the order of their joints (Scene.init reorders both)');
	end
	if isempty(j.parent)
		desc.parent(i) = -1;
	else
		p = 0;
		for k = 1 : i-1
			if scene.joints{k} == j.parent
				p = k;
			end
		end
		if p == 0
			error('redmax:hip','flattenScene: joint %d is listed before its parent',i);
		end
		desc.parent(i) = p - 1; % 0-based
	end
	t = find(strcmp(types(:,1),class(j)),1);
	if isempty(t)
		error('redmax:hip','flattenScene: joint class %s is not supported',class(j));
	end
	desc.type(i) = types{t,2};
	if isprop(j,'axis')
		desc.axis(:,i) = j.axis(:);
	end
	if isprop(j,'plane')
		desc.plane(:,i) = [j.plane(:,1); j.plane(:,2)];
	end
	if ~isempty(j.E0_pj)
		desc.E0_pj(:,:,i) = j.E0_pj;
	end
	desc.E0_ji(:,:,i) = j.body.E0_ji;
	desc.I_i(:,i) = j.body.I_i(:);
	if j.ndof > 0
		desc.qRest(i) = j.qRest(1);
		desc.qRestR(j.idxR) = j.qRest(:);
		if any(j.tau(:) ~= j.tau(1))
			error('redmax:hip','flattenScene: one torque value per joint is supported (joint %d has %d different ones)',i,j.ndof);
		end
		desc.tau(i) = j.tau(1);
	end
	desc.stiffness(i) = j.stiffness;
	desc.damping(i) = j.damping;
	desc.qLimL(i) = j.qLimL;
	desc.qLimU(i) = j.qLimU;
	desc.qLimK(i) = j.qLimK;
	desc.qLimD(i) = j.qLimD;
end

% scene.forces: ForceNull (nothing to do) or ForceGroundCuboid objects.  Every object holds its own E, kn, kt, mu, kd
% (ForceGroundCuboid.m:6-13): they travel per body (groundE_body 16 x n, kn_body .. kd_body 1 x n); groundE, kn .. kd are those
% of the first object (what a scene with one ground, like scene 11, needs).
ground = [];
for i = 1 : numel(scene.forces)
	f = scene.forces{i};
	switch class(f)
		case 'redmax.ForceNull'
		case 'redmax.ForceGroundCuboid'
			if isempty(ground)
				ground = f;
				desc.contact = zeros(1,n,'int32');
				desc.sides = zeros(3,n);
				desc.groundE = f.E;
				desc.kn = f.kn; desc.kt = f.kt; desc.mu = f.mu; desc.kd = f.kd;
				desc.groundE_body = repmat(reshape(f.E,16,1),1,n);
				desc.kn_body = f.kn*ones(1,n); desc.kt_body = f.kt*ones(1,n); desc.mu_body = f.mu*ones(1,n); desc.kd_body = f.kd*ones(1,n);
			end
			hit = 0;
			for k = 1 : n
				if scene.bodies{k} == f.cuboid
					hit = k;
				end
			end
			if hit == 0
				error('redmax:hip','flattenScene: ForceGroundCuboid on a body that is not in the scene');
			end
			if desc.contact(hit)
				% A second (third ...) ForceGroundCuboid on this cuboid - a floor and a wall; the reference's forces are a list
				% (Force.m:26-56) and nothing ties a body to one.  The library takes ONE force object per listing entry, so the force
				% is listed as a fixed, massless child of the body's joint with the body's own transform and sides: the same corners
				% moving with the same twist, hence the same wrench, K and D pulled through the same Jacobian rows.  No DOF is added
				% (redmax_amd/redmax.py Scene.desc does the same; tests/test_oracle_fd.py shows the two forms agree to roundoff).
				m = desc.njoints + 1;
				desc.njoints = m;
				desc.parent(m) = hit - 1; % 0-based: the joint of that body
				desc.type(m) = 0;         % RMX_JOINT_FIXED
				desc.axis(:,m) = [0;0;1];
				desc.E0_pj(:,:,m) = eye(4);
				desc.E0_ji(:,:,m) = f.cuboid.E0_ji;
				desc.I_i(:,m) = zeros(6,1);
				desc.qRest(m) = 0; desc.tau(m) = 0; desc.stiffness(m) = 0; desc.damping(m) = 0;
				desc.qLimL(m) = -inf; desc.qLimU(m) = inf; desc.qLimK(m) = 0; desc.qLimD(m) = 0;
				desc.plane(:,m) = [1;0;0;0;1;0];
				desc.contact(m) = 0;
				hit = m;
			end
			desc.contact(hit) = 1;
			desc.sides(:,hit) = f.cuboid.sides(:);
			desc.groundE_body(:,hit) = reshape(f.E,16,1);
			desc.kn_body(hit) = f.kn; desc.kt_body(hit) = f.kt; desc.mu_body(hit) = f.mu; desc.kd_body(hit) = f.kd;
		otherwise
			error('redmax:hip','flattenScene: force class %s is outside the HIP path (ForceNull and ForceGroundCuboid are in)',class(f));
	end
end
end
